// "Thin" conv1d kernels for single-input-channel layers with short kernels
// (AudioDiscriminator.l1, phase3/archis/default.py:298, and WaveGAN l1, :117:
// Conv1d(1, 32, 25, stride=4)). With Cin = 1 the contraction length is only k = 25, the
// arithmetic intensity is ~11 flop/B and the op is HBM-bound: the 157 MB
// activation / gradient tensor must be streamed once, everything else fits in cache. The
// GEMM engine's 128-wide tiles would be >80 % padding here; all three directions run on
// v_mfma_f32_32x32x2_f32 with the 32 channels as one tile dimension:
//   forward        : C[channel][position], K = taps, x windows gathered per lane, LDS transpose for 16-byte stores
//   backward-data  : Z[tap][position] = W^T dy (K = channels), then the overlap-add of the taps
//   backward-weight: C[tap][channel], K = positions (deterministic two-stage sum of per-block slabs, no atomics)
// Masks (activation derivative) follow the convention of gemm_engine.h: value *= (mask > 0 ? 1 : slope).
#include "m2d_common.h"

#define THIN_MAX_K 32

struct ThinArgs {
  const float* x;      // (B, 1, L)
  const float* w;      // (Cout, 1, ks)
  const float* bias;   // fwd only, may be NULL
  const float* dy;     // (B, Cout, Lout)   bwd
  const float* mask;   // optional
  float* out;          // fwd: y; bwd_data: dx; bwd_weight: partial slabs
  float* out2;         // bwd_weight: bias-gradient partial slabs (optional)
  float* stats;        // fwd (optional): per-wave partial (sum, sum of squares) per channel: [(block * 4 + wave)][32][2]
  int xT, xS, xhop;    // window view of a padded track: sample n starts at (n / xT) * xS + (n % xT) * xhop (xT = 0: dense)
  int B, L, Cout, ks, stride, pad, Lout;
  int act;
  float slope, mask_slope;
  int chunk;           // bwd_weight: positions per block
};

// Backward-weight on the matrix pipe for Cout == 32: dW^T[kk, co] = sum_l x[n, s*l - pad + kk] * dy[n, co, l]
// is a 32 x 32 (25 valid rows) tile accumulated over a very long K = l, i.e. one
// v_mfma_f32_32x32x2_f32 per two positions with NO cross-lane reduction. Each wave streams its own
// 64-position tiles: the (32 co x 64 l) slice of dy (times the activation mask) goes through a
// wave-private LDS image with coalesced 16-B loads, the x windows are read straight from
// global memory (lanes 0-31 read 32 consecutive samples). HBM-bound: dy (+ mask) is read once.
typedef float thin_f32x16 __attribute__((ext_vector_type(16)));
typedef float thin_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned thin_u32x4 __attribute__((ext_vector_type(4)));

template <int KS, int S, bool MASKED>
__global__ void __launch_bounds__(256) thin_bwd_weight_mfma_kernel(const ThinArgs a) {
  constexpr int LD = 65;             // [co][l] image, odd stride: fragment reads (lanes along co) are conflict-free
  constexpr int XW = 64 * S + 32;    // x window of one 64-position tile: positions S*lt - pad + [0, XW)
  constexpr int DEPTH = MASKED ? 1 : 2;  // dy tiles in flight per wave beyond the one being multiplied (the loop is
                                         // latency-bound: one 8 KB tile in flight per wave reached 1.9 TB/s)
  __shared__ float smem[4][32 * LD + XW];  // wave-private; reused for the cross-wave reduction at the end
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.y;
  const int l0 = blockIdx.x * a.chunk;
  int l1 = l0 + a.chunk;
  if (l1 > a.Lout) l1 = a.Lout;
  const float* xr = a.xT > 0 ? a.x + (size_t)(n / a.xT) * a.xS + (size_t)(n % a.xT) * a.xhop : a.x + (size_t)n * a.L;
  const float* dyn = a.dy + (size_t)n * 32 * a.Lout;
  const float* mkn = (MASKED && a.mask) ? a.mask + (size_t)n * 32 * a.Lout : nullptr;
  float* tl = smem[wave];
  float* xw = tl + 32 * LD;
  thin_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int i31 = lane & 31, h = lane >> 5;
  const int lq = 4 * (lane & 15);
  const bool vec = (a.Lout % 4) == 0, vec2 = (a.Lout % 2) == 0 && (l0 & 1) == 0;
  float4 dv[DEPTH][8], mv[MASKED ? 8 : 1];
  float xs[DEPTH][(XW + 63) / 64];
  // fetch(d, lt): this lane's 8 x float4 of the dy tile starting at position lt (and its mask), and its share of the
  // tile's x window; positions past the chunk / outside the sample read as zero
  auto fetch = [&](float4 (&d4)[8], float (&xq)[(XW + 63) / 64], int lt) {
#ifdef THIN_X_BW_NOLOAD   // experiment (wrong results): no global loads in the loop
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      d4[i] = make_float4((float)lt, 1.f, 2.f, (float)lane);
      if (MASKED) mv[i] = make_float4(1.f, -1.f, 1.f, (float)(lane - 32));
    }
#pragma unroll
    for (int i = 0; i < (XW + 63) / 64; ++i) xq[i] = (float)(lt + i);
    return;
#endif
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int co = (lane >> 4) + 4 * i;
      const size_t g = (size_t)co * a.Lout + lt + lq;
      if (vec && lt + lq + 3 < l1) {
        d4[i] = *reinterpret_cast<const float4*>(dyn + g);
        if (MASKED) mv[i] = mkn ? *reinterpret_cast<const float4*>(mkn + g) : make_float4(1.f, 1.f, 1.f, 1.f);
      } else if (vec2 && lt + lq + 3 < l1) {
        const float2 d0 = *reinterpret_cast<const float2*>(dyn + g);
        const float2 d1 = *reinterpret_cast<const float2*>(dyn + g + 2);
        d4[i] = make_float4(d0.x, d0.y, d1.x, d1.y);
        if (MASKED) {
          if (mkn) {
            const float2 m0 = *reinterpret_cast<const float2*>(mkn + g);
            const float2 m1 = *reinterpret_cast<const float2*>(mkn + g + 2);
            mv[i] = make_float4(m0.x, m0.y, m1.x, m1.y);
          } else {
            mv[i] = make_float4(1.f, 1.f, 1.f, 1.f);
          }
        }
      } else {
        float d[4], m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = lt + lq + j < l1;
          d[j] = ok ? dyn[g + j] : 0.f;
          m[j] = (MASKED && ok && mkn) ? mkn[g + j] : 1.f;
        }
        d4[i] = make_float4(d[0], d[1], d[2], d[3]);
        if (MASKED) mv[i] = make_float4(m[0], m[1], m[2], m[3]);
      }
    }
#pragma unroll
    for (int i = 0; i < (XW + 63) / 64; ++i) {
      const int e = lane + 64 * i;
      const int pos = lt * S - a.pad + e;
      xq[i] = (e < XW && lt < l1 && pos >= 0 && pos < a.L) ? xr[pos] : 0.f;
    }
  };
  const int lt0 = l0 + wave * 64;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (lt0 + 256 * d < l1) fetch(dv[d], xs[d], lt0 + 256 * d);
  for (int lt = lt0; lt < l1; lt += 256) {
    // land the oldest prefetched tile in the wave-private LDS image (mask applied here)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int co = (lane >> 4) + 4 * i;
      const float ms = a.mask_slope;
      tl[co * LD + lq + 0] = dv[0][i].x * ((!MASKED || mv[i].x > 0.f) ? 1.f : ms);
      tl[co * LD + lq + 1] = dv[0][i].y * ((!MASKED || mv[i].y > 0.f) ? 1.f : ms);
      tl[co * LD + lq + 2] = dv[0][i].z * ((!MASKED || mv[i].z > 0.f) ? 1.f : ms);
      tl[co * LD + lq + 3] = dv[0][i].w * ((!MASKED || mv[i].w > 0.f) ? 1.f : ms);
    }
#pragma unroll
    for (int i = 0; i < (XW + 63) / 64; ++i)
      if (lane + 64 * i < XW) xw[lane + 64 * i] = xs[0][i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // keep DEPTH tiles in flight under this tile's MFMAs
    if (DEPTH == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) dv[0][i] = dv[DEPTH - 1][i];
#pragma unroll
      for (int i = 0; i < (XW + 63) / 64; ++i) xs[0][i] = xs[DEPTH - 1][i];
    }
    if (lt + 256 * DEPTH < l1) fetch(dv[DEPTH - 1], xs[DEPTH - 1], lt + 256 * DEPTH);
    // A[kk = i31][k = h] = x[S * l - pad + kk], l = lt + 2 j + h: window element S * (2 j + h) + kk. Rows KS..30 hold
    // neighbouring samples (their outputs are never read); the spare row 31 = ones: C[31][co] = sum_l dy (bias gradient).
    // dy past the chunk was staged as zero, so no position test is needed here.
    const float* xa = xw + S * h + i31;
    const float* bb = tl + i31 * LD + h;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
      float av = xa[2 * S * j];
      if (KS < 32) av = i31 == 31 ? 1.f : av;
#ifdef THIN_X_BW_NOMFMA   // experiment (wrong results)
      acc[j & 15] += av * bb[2 * j];
#else
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bb[2 * j], acc, 0, 0, 0);
#endif
    }
    __builtin_amdgcn_wave_barrier();
  }
  // C[i = kk][j = co]: col = lane & 31 (co), row = (r & 3) + 8 * (r >> 2) + 4 * h (kk)
  __syncthreads();
  float* red = &smem[0][0];  // [4][32 * 32] fits: 4 * (32 * LD + XW) floats
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i31] = acc[r];
  __syncthreads();
  const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  for (int o = threadIdx.x; o < 32 * KS; o += 256) {
    const int co = o / KS, kk = o % KS;
    const int idx = kk * 32 + co;
    a.out[blk * 32 * KS + o] = red[idx] + red[1024 + idx] + red[2048 + idx] + red[3072 + idx];
  }
  if (a.out2 && threadIdx.x < 32) {
    const int idx = 31 * 32 + threadIdx.x;
    a.out2[blk * 32 + threadIdx.x] = red[idx] + red[1024 + idx] + red[2048 + idx] + red[3072 + idx];
  }
}

// Backward-data on the matrix pipe (KS = 25, S = 4, Cout = 32): per position l the 32 x 25 product
//   Z[l][kk] = sum_co dy[n, co, l] * W[co, kk]                  (one 32x32 MFMA tile per 32 positions, K = co)
// followed by the overlap-add  dx[n, S*q + rho - pad] = sum_t Z[q - t][rho + S*t]  (rho + S*t < KS).
// A wave walks a run of positions in 64-wide tiles: the dy tile goes through a wave-private LDS image (coalesced
// 16-byte loads, mask applied on the way in, one or two tiles in flight as in backward-weight), Z comes back
// through the same image as [position][tap] rows (stride 33: lanes along positions are conflict-free), each lane
// sums the <= 7 x 4 taps of its q and the 256 outputs of the tile leave as four coalesced dword stores. The six Z rows
// a tile needs from its predecessor travel in three registers per lane; a run starts with one 32-position tile for them.
template <int KS, int S, bool MASKED>
__global__ void __launch_bounds__(256) thin_bwd_data_mfma_kernel(const ThinArgs a) {
  constexpr int LD = 65;   // dy image [co][64 positions]
  constexpr int ZLD = 33;  // Z image [6 + 64 positions][32 taps]
  constexpr int T = (KS + S - 1) / S;  // 7
  constexpr int HALO = T - 1;          // 6
  constexpr int DEPTH = MASKED ? 1 : 2;
  __shared__ float smem[4][(64 + HALO) * ZLD];  // >= 32 * LD
  static_assert((64 + HALO) * ZLD >= 32 * LD, "image size");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int n = blockIdx.y;
  const int nq = (a.L - 1 + a.pad) / S + 1;  // q in [0, nq)
  const int run = a.chunk / 4;               // positions per wave (multiple of 64)
  const int qa = blockIdx.x * a.chunk + wave * run;
  int qb = qa + run;
  if (qb > nq) qb = nq;
  if (qa >= nq) return;
  const float* dyn = a.dy + (size_t)n * 32 * a.Lout;
  const float* mkn = (MASKED && a.mask) ? a.mask + (size_t)n * 32 * a.Lout : nullptr;
  float* dxn = a.out + (size_t)n * a.L;
  float* buf = smem[wave];
  float wa[16];  // A[row = kk][k = co]: W[co][kk], co = 2 s + h
#pragma unroll
  for (int s2 = 0; s2 < 16; ++s2) wa[s2] = l31 < KS ? a.w[(2 * s2 + h) * KS + l31] : 0.f;
  const int lq = 4 * (lane & 15);
  const bool vec = (a.Lout % 4) == 0, vec2 = (a.Lout % 2) == 0;
  float4 dv[DEPTH][8], mv[MASKED ? 8 : 1];
  // this lane's 8 x float4 of the dy tile [32 co][64 positions from lt]; positions outside [0, Lout) read as zero
  auto fetch = [&](float4 (&d4)[8], int lt) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int co = (lane >> 4) + 4 * i;
      const int l = lt + lq;
      const size_t g = (size_t)co * a.Lout + l;
      if (vec && l >= 0 && l + 3 < a.Lout) {
        d4[i] = *reinterpret_cast<const float4*>(dyn + g);
        if (MASKED) mv[i] = mkn ? *reinterpret_cast<const float4*>(mkn + g) : make_float4(1.f, 1.f, 1.f, 1.f);
      } else if (vec2 && l >= 0 && l + 3 < a.Lout) {
        const float2 d0 = *reinterpret_cast<const float2*>(dyn + g);
        const float2 d1 = *reinterpret_cast<const float2*>(dyn + g + 2);
        d4[i] = make_float4(d0.x, d0.y, d1.x, d1.y);
        if (MASKED) {
          if (mkn) {
            const float2 m0 = *reinterpret_cast<const float2*>(mkn + g);
            const float2 m1 = *reinterpret_cast<const float2*>(mkn + g + 2);
            mv[i] = make_float4(m0.x, m0.y, m1.x, m1.y);
          } else {
            mv[i] = make_float4(1.f, 1.f, 1.f, 1.f);
          }
        }
      } else {
        float d[4], m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = l + j >= 0 && l + j < a.Lout;
          d[j] = ok ? dyn[(ptrdiff_t)g + j] : 0.f;
          m[j] = (MASKED && ok && mkn) ? mkn[(ptrdiff_t)g + j] : 1.f;
        }
        d4[i] = make_float4(d[0], d[1], d[2], d[3]);
        if (MASKED) mv[i] = make_float4(m[0], m[1], m[2], m[3]);
      }
    }
  };
  // registers -> dy image (mask applied)
  auto land = [&](const float4 (&d4)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int co = (lane >> 4) + 4 * i;
      const float ms = a.mask_slope;
      buf[co * LD + lq + 0] = d4[i].x * ((!MASKED || mv[i].x > 0.f) ? 1.f : ms);
      buf[co * LD + lq + 1] = d4[i].y * ((!MASKED || mv[i].y > 0.f) ? 1.f : ms);
      buf[co * LD + lq + 2] = d4[i].z * ((!MASKED || mv[i].z > 0.f) ? 1.f : ms);
      buf[co * LD + lq + 3] = d4[i].w * ((!MASKED || mv[i].w > 0.f) ? 1.f : ms);
    }
  };
  auto sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // Z tile of 32 positions starting at image column c0: D[row = kk][col = position]
  auto zmma = [&](int c0) {
    thin_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s2], buf[(2 * s2 + h) * LD + c0 + l31], acc, 0, 0, 0);
    return acc;
  };
  // carry: Z rows of the HALO positions before the tile, element e = lane + 64 k -> (row e / 32, tap e % 32)
  float carry[3] = {0.f, 0.f, 0.f};
  if (qa > 0) {
    // positions [qa - 32, qa): only the last HALO columns are kept
    fetch(dv[0], qa - 64);  // the 64-wide fetch ending at qa; columns 32..63 are the ones multiplied
    land(dv[0]);
    sync();
    const thin_f32x16 z = zmma(32);
    sync();
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (l31 >= 32 - HALO) buf[(l31 - (32 - HALO)) * ZLD + (r & 3) + 8 * (r >> 2) + 4 * h] = z[r];
    sync();
#pragma unroll
    for (int k = 0; k < 3; ++k) carry[k] = buf[((lane + 64 * k) >> 5) * ZLD + ((lane + 64 * k) & 31)];
    sync();
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (qa + 64 * d < qb) fetch(dv[d], qa + 64 * d);
  for (int l0 = qa; l0 < qb; l0 += 64) {
    land(dv[0]);
    sync();
    if (DEPTH == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) dv[0][i] = dv[1][i];
    }
    if (l0 + 64 * DEPTH < qb) fetch(dv[DEPTH - 1], l0 + 64 * DEPTH);
    const thin_f32x16 z0 = zmma(0), z1 = zmma(32);
    sync();
    // Z image: rows 0..HALO-1 from the carry, rows HALO.. = this tile's 64 positions
#pragma unroll
    for (int k = 0; k < 3; ++k) buf[((lane + 64 * k) >> 5) * ZLD + ((lane + 64 * k) & 31)] = carry[k];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = (r & 3) + 8 * (r >> 2) + 4 * h;
      buf[(HALO + l31) * ZLD + kk] = z0[r];
      buf[(HALO + 32 + l31) * ZLD + kk] = z1[r];
    }
    sync();
    // lane = q - l0: dx[S q + rho - pad] = sum_t Z[q - t][rho + S t]
    float o[S];
#pragma unroll
    for (int rho = 0; rho < S; ++rho) {
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t)
        if (rho + S * t < KS) acc += buf[(HALO + lane - t) * ZLD + rho + S * t];
      o[rho] = acc;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) carry[k] = buf[(64 + ((lane + 64 * k) >> 5)) * ZLD + ((lane + 64 * k) & 31)];
    sync();
#pragma unroll
    for (int rho = 0; rho < S; ++rho) buf[S * lane + rho] = o[rho];
    sync();
    const int j0 = S * l0 - a.pad;
#pragma unroll
    for (int k = 0; k < S; ++k) {
      const int e = lane + 64 * k;  // output S * (l0 + e / S) + e % S - pad, q = l0 + e / S < qb
      const int j = j0 + e;
      if (j >= 0 && j < a.L && l0 + e / S < qb) dxn[j] = buf[e];
    }
    sync();
  }
}

// dw[o] = sum over the nblk partial slabs, fixed order: 32 groups of slabs per output (group g takes
// slabs g, g + 32, ...) summed in fp64, then combined in group order. A block owns 8 outputs, so a
// 32 x 25 weight gradient is summed by 100 blocks (one chain over all slabs per thread was latency-bound:
// 77 us for 320 slabs; 13 blocks of 16 groups: 24 us for 1 216 slabs).
__global__ void __launch_bounds__(256) thin_sum_partials_kernel(const float* partial, float* dw, int nblk, int n_out,
                                                                  const float* partial2, float* out2, int n_out2) {
  __shared__ double part[32][8];
  const int ox = threadIdx.x & 7, g = threadIdx.x >> 3;
  const int nb1 = (n_out + 7) / 8;
  int blk = blockIdx.x;
  if (blk >= nb1) {  // the trailing blocks sum the bias partials the same way
    blk -= nb1;
    partial = partial2;
    dw = out2;
    n_out = n_out2;
  }
  const int o = blk * 8 + ox;
  double s = 0.0;
  if (o < n_out)
    for (int b = g; b < nblk; b += 32) s += (double)partial[(size_t)b * n_out + o];
  part[g][ox] = s;
  __syncthreads();
  if (g == 0 && o < n_out) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 32; ++j) t += part[j][ox];
    dw[o] = (float)t;
  }
}

static bool thin_ok(int Cin, int Cout, int ks, int stride) {
  // the layers that exist: AudioDiscriminator.l1 / WaveGANAudioEncoder.l1, Conv1d(1, 32, 25, stride=4)
  return Cin == 1 && ks == 25 && stride == 4 && Cout == 32;
}

bool m2d_thin_applicable(int Cin, int Cout, int ks, int stride) { return thin_ok(Cin, Cout, ks, stride); }

// backward-weight: positions per block. About two blocks per CU of work (512 blocks: measured best at
// B = 64, Lout = 19 200: 52 us at 2 560 positions against 65 us at 1 024 and 54 us at 4 864), whole 256-position
// rounds of the block's four waves.
static int thin_bw_chunk(int B, int Lout) {
  const long long per = ((long long)B * Lout + 511) / 512;
  long long c = (per + 255) / 256 * 256;
  if (c < 1024) c = 1024;
  if (c > 8192) c = 8192;
  return (int)c;
}

size_t m2d_thin_bwd_weight_ws(int B, int Cout, int ks, int Lout) {
  const size_t nblk = (size_t)B * m2d_ceil_div(Lout, thin_bw_chunk(B, Lout));
  return nblk * (Cout * ks + Cout) * sizeof(float);  // weight partials + bias partials
}

// Forward for Cout == 32 on the matrix pipe: a tile = 32 channels x 32 consecutive positions of one sample = one 32x32
// accumulator of v_mfma_f32_32x32x2_f32, K = the k taps (13 steps for k = 25; the weights stay in 13 registers per lane
// for the whole launch). 13 MFMAs per 4 KB of output instead of 25 vector FMAs per output.
//   A[row = lane & 31][k = 2*ks + (lane >> 5)] = W[row, k]
//   B[k][col = lane & 31]                      = x[n, (l0 + col)*S - pad + k]
//   C: col = lane & 31 (position), row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5) (channel)
// The accumulators go through a wave-private LDS image [channel][32 positions] so that the epilogue reads the mask and
// writes the result as dwordx4 runs of 128 bytes per channel row (measured at B = 64: 79 us plain / 107 us masked,
// against 71 / 161 us for dword accesses straight from the accumulator layout, and ~117 us for the vector-ALU kernel).
// Round 6: PERSISTENT waves. With one tile per wave (38 400 waves at B = 64) the launch took 68 us plain / 87 us masked,
// and 39 / 44 us of that with neither the gathers nor the stores compiled in (tools/thin_time.py, -DTHIN_X_NOGATHER
// -DTHIN_X_NOSTORE): every wave paid the weight loads, the gather latency and its own store drain one after the other.
// Now the grid is one full-occupancy round (8 workgroups per CU); a wave walks the tiles g, g + G, g + 2G, ... (G = waves
// of the grid; the four waves of a workgroup hold four neighbouring tiles: 512 contiguous bytes per channel row) and
// keeps the NEXT tile's 13 gathered x values (and this tile's mask rows) in flight under this tile's MFMAs, LDS round
// trip and stores.
// Instruction budget (the launch is ISSUE-bound, not HBM-bound: with neither gathers nor stores compiled in it still took
// 39 of its 68 us): 13 MFMAs = 832 matrix-pipe cycles per tile, and the first version spent more than that again on
// vector-ALU work per tile - per-element bounds selects of the gathers, 64-bit addresses, bias loads. Now: the gathers are
// raw buffer loads off a per-sample descriptor (positions left of the sample wrap to huge offsets, positions right of it
// exceed num_records: the range check returns 0.0, no selects), interior tiles (all 32 positions inside the row, 16-byte
// aligned rows) take a pass with 32-bit offsets off a per-sample output descriptor and no per-element tests; the bias
// values of a lane's four channel rows are loaded once per launch.
template <int KS, int S>
#ifndef THIN_WPE
#define THIN_WPE 4
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(THIN_WPE, 8)))
thin_fwd_mfma_kernel(const ThinArgs a, int tiles_per_row, int total_tiles) {
  constexpr int NS = (KS + 1) / 2;
  constexpr int LDP = 32 + 4;
  __shared__ float img[4][32 * LDP];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c31 = lane & 31, h = lane >> 5;
  float* im = img[wave];
  float wa[NS];
#pragma unroll
  for (int ks = 0; ks < NS; ++ks) {
    const int k = 2 * ks + h;
    wa[ks] = k < KS ? a.w[c31 * KS + k] : 0.f;
  }
  const int nwaves = gridDim.x * 4;
  // Row starts are Lout floats apart: 16-byte stores need Lout % 4 == 0, 8-byte ones Lout % 2 == 0
  // (WaveGAN's 794-position rows take the float2 path; odd lengths store scalars).
  const int valign = (a.Lout % 4 == 0) ? 4 : (a.Lout % 2 == 0) ? 2 : 1;
  const int cq = lane & 7;          // 8 lanes per channel row in the epilogue, 8 rows per store instruction
  const int crow = lane >> 3;
  float bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = a.bias ? a.bias[crow + 8 * i] : 0.f;
  const float act_s = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : a.slope);   // max(v, 0) + s min(v, 0)
  const float ms = a.mask_slope;
  double st1[4] = {0.0, 0.0, 0.0, 0.0}, st2[4] = {0.0, 0.0, 0.0, 0.0};   // statistics of this wave's tiles (a lane's share)
  const unsigned row_bytes = (unsigned)a.Lout * 4u;
  const unsigned sample_bytes = 32u * row_bytes;   // (launcher: 32 Lout floats < 2^29)
  // the odd half's last tap pair reaches k = KS (odd KS): its weight is 0.0, but 0 x inf = NaN - that one operand is forced to 0
  constexpr bool LAST_DEAD = (KS & 1) != 0;

  auto fetch = [&](int tile, float (&dst)[NS]) {
    const int n = tile / tiles_per_row;
    const int p0 = (tile - n * tiles_per_row) * 32;
    const float* xr = a.xT > 0 ? a.x + (size_t)(n / a.xT) * a.xS + (size_t)(n % a.xT) * a.xhop : a.x + (size_t)n * a.L;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xr, (short)0, (int)((unsigned)a.L * 4u), 0x00020000);
    const int l = p0 + c31;
    // (positions past the row's end: l >= Lout -> force out of range; their products are never stored)
    const unsigned voff = l < a.Lout ? (unsigned)((l * S - a.pad + h) * 4) : 0x80000000u;
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
#ifdef THIN_X_NOGATHER   // experiment (wrong results): no x gathers
      dst[ks] = (float)((voff + ks) & 7);
#else
      dst[ks] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)voff + 8 * ks, 0, 0));
#endif
    }
    if (LAST_DEAD && h == 1) dst[NS - 1] = 0.f;
  };
  auto process = [&](int tile, const float (&xb)[NS]) {
    const int n = tile / tiles_per_row;
    const int p0 = (tile - n * tiles_per_row) * 32;
    const bool interior = valign == 4 && p0 + 32 <= a.Lout;   // (uniform)
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)n * 32 * a.Lout), (short)0, (int)sample_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)((a.mask ? a.mask : a.out) + (size_t)n * 32 * a.Lout), (short)0,
                                                                        (int)(a.mask ? sample_bytes : 0u), 0x00020000);
    const unsigned eo = (unsigned)crow * row_bytes + (unsigned)(p0 + 4 * cq) * 4u;   // channel row crow + 8 i: + 8 i row_bytes
    // this tile's mask rows, fetched before the MFMAs (interior tiles; the other path reads them in place)
    thin_f32x4 mk[4];
    if (interior && a.mask) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        mk[i] = __builtin_bit_cast(thin_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, (int)(eo + (unsigned)(8 * i) * row_bytes), 0, 0));
    }
    thin_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ks], xb[ks], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) im[((r & 3) + 8 * (r >> 2) + 4 * h) * LDP + c31] = acc[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (interior) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = crow + 8 * i;
        const thin_f32x4 q = *reinterpret_cast<const thin_f32x4*>(im + co * LDP + 4 * cq);
        thin_f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float t = q[j] + bv[i];
          v[j] = fmaxf(t, 0.f) + act_s * fminf(t, 0.f);
        }
        if (a.mask) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] *= mk[i][j] > 0.f ? 1.f : ms;
        }
#ifdef THIN_X_NOSTORE    // experiment (wrong results): no output stores
        if (v[0] == 12345.678f)
#endif
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(thin_u32x4, v), ro, (int)(eo + (unsigned)(8 * i) * row_bytes), 0, 0);
        // statistics: summed per lane over ALL the wave's tiles; the 8 lanes of a channel row meet once, at the end
        st1[i] += (v[0] + v[1]) + (v[2] + v[3]);
        st2[i] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      }
    } else {
      // the last tile of a row, or rows that are not 16-byte aligned: per-element tests, 8-byte / scalar accesses
      const bool whole = p0 + 4 * cq + 4 <= a.Lout;
#pragma unroll 1
      for (int i = 0; i < 4; ++i) {
        const int co = crow + 8 * i;
        const float4 q = *reinterpret_cast<const float4*>(im + co * LDP + 4 * cq);
        const float b_ = i == 0 ? bv[0] : (i == 1 ? bv[1] : (i == 2 ? bv[2] : bv[3]));
        float v[4] = {q.x + b_, q.y + b_, q.z + b_, q.w + b_};
        const size_t o = ((size_t)n * 32 + co) * a.Lout + p0 + 4 * cq;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f) + act_s * fminf(v[j], 0.f);
        if (whole && valign == 4) {
          if (a.mask) {
            const float4 m = *reinterpret_cast<const float4*>(a.mask + o);
            v[0] *= m.x > 0.f ? 1.f : ms;
            v[1] *= m.y > 0.f ? 1.f : ms;
            v[2] *= m.z > 0.f ? 1.f : ms;
            v[3] *= m.w > 0.f ? 1.f : ms;
          }
          *reinterpret_cast<float4*>(a.out + o) = make_float4(v[0], v[1], v[2], v[3]);
        } else if (whole && valign == 2) {
          if (a.mask) {
            const float2 m0 = *reinterpret_cast<const float2*>(a.mask + o);
            const float2 m1 = *reinterpret_cast<const float2*>(a.mask + o + 2);
            v[0] *= m0.x > 0.f ? 1.f : ms;
            v[1] *= m0.y > 0.f ? 1.f : ms;
            v[2] *= m1.x > 0.f ? 1.f : ms;
            v[3] *= m1.y > 0.f ? 1.f : ms;
          }
          *reinterpret_cast<float2*>(a.out + o) = make_float2(v[0], v[1]);
          *reinterpret_cast<float2*>(a.out + o + 2) = make_float2(v[2], v[3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (p0 + 4 * cq + j < a.Lout) {
              float y = v[j];
              if (a.mask) y *= a.mask[o + j] > 0.f ? 1.f : ms;
              a.out[o + j] = y;
              v[j] = y;
            } else {
              v[j] = 0.f;
            }
          }
        }
        {
          const float t1 = (v[0] + v[1]) + (v[2] + v[3]);
          const float t2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
          if (i == 0) { st1[0] += t1; st2[0] += t2; }
          else if (i == 1) { st1[1] += t1; st2[1] += t2; }
          else if (i == 2) { st1[2] += t1; st2[2] += t2; }
          else { st1[3] += t1; st2[3] += t2; }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // the image is read before the next tile overwrites it (one wave, in order)
  };

  int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  const int gw = tile;   // this wave's slot of the statistics partials
  auto flush_stats = [&]() {
    if (!a.stats) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // (fp64 across the wave's tiles and lanes - a tile's four values per lane are summed in fp32 as before; the
      // partial is rounded to fp32 once, when it is written: as accurate as the per-tile partials it replaces)
      double s1 = st1[i], s2 = st2[i];
#pragma unroll
      for (int off = 4; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off, 64);
        s2 += __shfl_xor(s2, off, 64);
      }
      if (cq == 0) {
        float* dst = a.stats + (size_t)gw * 64 + 2 * (crow + 8 * i);
        dst[0] = (float)s1;
        dst[1] = (float)s2;
      }
    }
  };
  if (tile >= total_tiles) {   // (a wave without a tile still owns a slot: zeros)
    flush_stats();
    return;
  }
  float xa[NS], xb2[NS];
  fetch(tile, xa);
#pragma unroll 1
  for (;;) {
    const int t2 = tile + nwaves;
    if (t2 < total_tiles) fetch(t2, xb2);
    process(tile, xa);
    if (t2 >= total_tiles) break;
    const int t3 = t2 + nwaves;
    if (t3 < total_tiles) fetch(t3, xa);
    process(t2, xb2);
    if (t3 >= total_tiles) break;
    tile = t3;
  }
  flush_stats();
}

struct M2dWinView {
  int T, S, hop;
};

int m2d_rowsums_reduce(const float* part, int P, int M, double* sums, double* scratch, hipStream_t stream);  // gemm_engine.hip

// per-tile partials + the scratch of the two-stage row sum (256 groups x 32 rows x fp64 pair)
static int thin_fwd_tiles_per_row(int Lout) { return m2d_ceil_div(Lout, 32); }
// statistics partials: one (sum, sum of squares) pair per channel and WAVE of the grid (at most 256 x 16 workgroups x 4)
#define THIN_FWD_MAX_WAVES (256 * 16 * 4)
static size_t thin_fwd_stats_part(int B, int Lout) { (void)B; (void)Lout; return (size_t)THIN_FWD_MAX_WAVES * 64 * sizeof(float); }
size_t m2d_thin_fwd_stats_ws(int B, int Lout) { return thin_fwd_stats_part(B, Lout) + (size_t)256 * 32 * 2 * sizeof(double); }
// workgroups of the persistent forward: one full-occupancy round (8 per CU: 18 KB of LDS, 35-50 registers), fewer when
// there are not that many groups of four tiles. M2D_THIN_WG_PER_CU: A/B lever
static int thin_fwd_grid(int total_tiles) {
  static const int per_cu = [] { const char* e = getenv("M2D_THIN_WG_PER_CU"); const int v = e ? atoi(e) : THIN_WPE; return v >= 1 && v <= 16 ? v : THIN_WPE; }();
  const int need = m2d_ceil_div(total_tiles, 4), full = 256 * per_cu;
  return need < full ? need : full;
}

int m2d_thin_fwd(const float* x, const float* w, const float* bias, float* y, int B, int L, int Cout, int ks,
                 int stride, int pad, int Lout, int act, float slope, const float* out_mask, float out_mask_slope,
                 const M2dWinView* wv, double* stats, void* ws, size_t ws_bytes, hipStream_t stream) {
  ThinArgs a;
  memset(&a, 0, sizeof(a));
  if (wv) { a.xT = wv->T; a.xS = wv->S; a.xhop = wv->hop; }
  const int tpr = thin_fwd_tiles_per_row(Lout);
  if ((long long)B * tpr >= (1LL << 30) || (long long)Lout * 32 * 4 >= (1LL << 31) || (long long)L * 4 >= (1LL << 31))
    M2D_FAIL(M2D_ERR_RANGE, "m2d_conv1d_fwd (thin): too many tiles / a sample beyond 2 GiB");
  const int total = B * tpr;
  if (stats) {   // (every wave of the grid writes its slot: nothing to zero)
    if (!ws || ws_bytes < m2d_thin_fwd_stats_ws(B, Lout)) M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_conv1d_fwd (thin): no room for the statistics partials");
    a.stats = (float*)ws;
  }
  a.x = x; a.w = w; a.bias = bias; a.mask = out_mask; a.out = y;
  a.B = B; a.L = L; a.Cout = Cout; a.ks = ks; a.stride = stride; a.pad = pad; a.Lout = Lout;
  a.act = act; a.slope = slope; a.mask_slope = out_mask_slope;
  M2dProfScope prof(M2D_FAM_POINTWISE, stream, 2.0 * B * Lout * (double)Cout * ks,
                    4.0 * B * ((double)L + (double)Cout * Lout * (out_mask ? 2 : 1)), "thin_conv_fwd", Cout, B * Lout, ks);
  const int grid = thin_fwd_grid(total);
  hipLaunchKernelGGL((thin_fwd_mfma_kernel<25, 4>), dim3(grid), dim3(256), 0, stream, a, tpr, total);
  M2D_CHECK_LAUNCH("thin_fwd_mfma_kernel");
  if (stats)
    return m2d_rowsums_reduce(a.stats, grid * 4, 32, stats, (double*)((char*)a.stats + thin_fwd_stats_part(B, Lout)), stream);
  return M2D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The audio encoder's first conv, Conv1d(1, 32, 250, stride 50, pad 124) on 3 200-sample windows
// (phase3/archis/default.py:64 of the reference; DefaultAudioEncoder.conv_layers[0]): Cin = 1 again, but K = 250 taps.
// On the engine it is a 32-row GEMM with N = 491 520 at B = 64 (the 32-row tiles: 9.4 scalar instructions per MFMA,
// MFMA busy 0.60): 163 us = 48 TFLOP/s in the step. Here a tile = 32 channels x 32 positions x all 250 taps = 125
// v_mfma_f32_32x32x2_f32 in a row on one accumulator:
//   A[row = channel c31][k = 2 ks + h] = W[c31][k]           from an LDS image of the weights [k][32] (block-shared, 32 KB)
//   B[k][col = position c31]           = x[(p0 + c31) S - pad + k]   from the tile's x segment in LDS: (32 - 1) S + KS = 1 800
//                                        consecutive samples of the window, zeros outside it (range-checked buffer loads)
// The 125 k-steps are unrolled: every LDS offset is an instruction immediate. Persistent waves (8 per workgroup, one
// workgroup per CU) walk the tiles g, g + G, ...; the NEXT tile's segment is fetched into registers (7 x 16 bytes per lane)
// before this tile's MFMAs and written to LDS after them. Epilogue as the k25 kernel's interior pass: wave-private LDS image,
// 16-byte stores (a window's 32 x 64 outputs are 8 KB contiguous), statistics summed per wave over all its tiles.
template <int KS, int S>
__global__ void __launch_bounds__(512) thin_long_fwd_kernel(const ThinArgs a, int tiles_per_row, int total_tiles) {
  static_assert(KS % 2 == 0 && S % 2 == 0, "even taps / stride");
  constexpr int NS = KS / 2;
  constexpr int SEG = 31 * S + KS + 3;         // samples a tile reads (+ up to 3 of lead-in: the segment starts on the
                                               // window's 16-byte grid, `lead` = pad rounded up to 4, minus pad)
  constexpr int SEG4 = (SEG + 3) / 4;          // 16-byte pieces
  constexpr int NP = (SEG4 + 63) / 64;         // pieces per lane
  constexpr int LDP = 32 + 4;
  constexpr int WAVE_FLOATS = SEG4 * 4 + 32 * LDP;
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  float* Ws = lsm;                             // [KS][32]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* xs = lsm + KS * 32 + wave * WAVE_FLOATS;
  float* im = xs + SEG4 * 4;
  const int c31 = lane & 31, h = lane >> 5;
  const int lead = ((a.pad + 3) & ~3) - a.pad;
  for (int e = threadIdx.x; e < KS * 32; e += 512) {
    const int k = e >> 5, c = e & 31;
    Ws[e] = a.w[c * KS + k];
  }
  __syncthreads();
  const int nwaves = gridDim.x * 8;
  const int cq = lane & 7, crow = lane >> 3;
  float bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = a.bias ? a.bias[crow + 8 * i] : 0.f;
  const float act_s = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : a.slope);
  const unsigned row_bytes = (unsigned)a.Lout * 4u;
  const unsigned sample_bytes = 32u * row_bytes;

  thin_f32x4 seg[NP];
  auto fetch = [&](int tile) {
    const int n = tile / tiles_per_row;
    const int p0 = (tile - n * tiles_per_row) * 32;
    const float* xr = a.xT > 0 ? a.x + (size_t)(n / a.xT) * a.xS + (size_t)(n % a.xT) * a.xhop : a.x + (size_t)n * a.L;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xr, (short)0, (int)((unsigned)a.L * 4u), 0x00020000);
    // piece q of the segment = samples p0 S - pad4 + 4 q ..+3, pad4 = pad rounded up to 4 (S, L multiples of 4: a piece
    // is inside the window or outside it)
    const int first = p0 * S - a.pad - lead;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int q = lane + 64 * i;
      const unsigned voff = q < SEG4 ? (unsigned)((first + 4 * q) * 4) : 0x80000000u;
      seg[i] = __builtin_bit_cast(thin_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)voff, 0, 0));
    }
  };
  auto land = [&]() {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int q = lane + 64 * i;
      if (q < SEG4) *reinterpret_cast<thin_f32x4*>(xs + 4 * q) = seg[i];
    }
  };
  int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * 8 + wave);
  const int gw = tile;   // this wave's slot of the statistics partials
  double st1[4] = {0.0, 0.0, 0.0, 0.0}, st2[4] = {0.0, 0.0, 0.0, 0.0};
  auto flush_stats = [&]() {
    if (!a.stats) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // (fp64 across the wave's tiles and lanes - a tile's four values per lane are summed in fp32 as before; the
      // partial is rounded to fp32 once, when it is written: as accurate as the per-tile partials it replaces)
      double s1 = st1[i], s2 = st2[i];
#pragma unroll
      for (int off = 4; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off, 64);
        s2 += __shfl_xor(s2, off, 64);
      }
      if (cq == 0) {
        float* dst = a.stats + (size_t)gw * 64 + 2 * (crow + 8 * i);
        dst[0] = (float)s1;
        dst[1] = (float)s2;
      }
    }
  };
  if (tile >= total_tiles) {   // (a wave without a tile still owns a slot: zeros)
    flush_stats();
    return;
  }
  fetch(tile);
  land();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float* ap = Ws + h * 32 + c31;
  const float* bp = xs + c31 * S + h + lead;
#pragma unroll 1
  for (;;) {
    const int next = tile + nwaves;
    if (next < total_tiles) fetch(next);
    thin_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[ks * 64], bp[2 * ks], acc, 0, 0, 0);
    const int n = tile / tiles_per_row;
    const int p0 = (tile - n * tiles_per_row) * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) im[((r & 3) + 8 * (r >> 2) + 4 * h) * LDP + c31] = acc[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (the segment's reads are done: the next one may land while the epilogue runs)
    if (next < total_tiles) land();
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)n * 32 * a.Lout), (short)0, (int)sample_bytes, 0x00020000);
    const unsigned eo = (unsigned)crow * row_bytes + (unsigned)(p0 + 4 * cq) * 4u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = crow + 8 * i;
      const thin_f32x4 q = *reinterpret_cast<const thin_f32x4*>(im + co * LDP + 4 * cq);
      thin_f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = q[j] + bv[i];
        v[j] = fmaxf(t, 0.f) + act_s * fminf(t, 0.f);
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(thin_u32x4, v), ro, (int)(eo + (unsigned)(8 * i) * row_bytes), 0, 0);
      // statistics: a lane's four values of channel row crow + 8 i, summed over ALL the wave's tiles in registers; the
      // eight lanes of a row meet once, after the last tile (one partial per (wave, channel): the tile -> wave assignment
      // is fixed, so the sums are repeatable)
      st1[i] += (v[0] + v[1]) + (v[2] + v[3]);
      st2[i] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // image read, next segment landed: visible to the next tile's fragment reads
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (next >= total_tiles) break;
    tile = next;
  }
  flush_stats();
}

static bool thin_long_enabled() {
  static const bool on = [] { const char* e = getenv("M2D_THIN_LONG"); return !(e && e[0] == '0'); }();   // A/B lever
  return on;
}
// Conv1d(1, 32, 250, 50, 124) (DefaultAudioEncoder) and Conv1d(1, 32, 160, 4, 79) (UNetAudioEncoder, phase3/archis/
// default.py:210) on windows whose length is a multiple of 4 and whose output rows are whole 32-position tiles
static int thin_long_kind(int Cin, int Cout, int ks, int stride) {
  if (Cin != 1 || Cout != 32) return 0;
  if (ks == 250 && stride == 50) return 1;
  if (ks == 160 && stride == 4) return 2;
  return 0;
}
bool m2d_thin_long_applicable(int Cin, int Cout, int ks, int stride, int pad, int L) {
  if (!thin_long_enabled() || !thin_long_kind(Cin, Cout, ks, stride)) return false;
  if ((L & 3) || pad < 0 || pad >= ks) return false;
  const int Lout = (L + 2 * pad - ks) / stride + 1;
  return Lout > 0 && Lout % 32 == 0;
}
// statistics partials: one (sum, sum of squares) pair per channel and WAVE of the grid (at most 512 workgroups x 8)
#define THIN_LONG_MAX_WAVES (512 * 8)
size_t m2d_thin_long_stats_ws(int B, int Lout) { (void)B; (void)Lout; return (size_t)THIN_LONG_MAX_WAVES * 64 * sizeof(float) + (size_t)256 * 32 * 2 * sizeof(double); }

template <int KS, int S>
static int thin_long_launch(const ThinArgs& a, int tpr, int total, hipStream_t stream, int* waves) {
  constexpr int SEG4 = (31 * S + KS + 3 + 3) / 4;
  constexpr size_t lds = (size_t)(KS * 32 + 8 * (SEG4 * 4 + 32 * 36)) * sizeof(float);
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_long_fwd_kernel<KS, S>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
  if (!attr_ok) M2D_FAIL(M2D_ERR_HIP, "m2d_conv1d_fwd (long single-channel kernel): cannot reserve %zu bytes of LDS", lds);
  // one workgroup (8 waves) per CU, two when the LDS image is small enough for two to be resident
  const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
  const int need = m2d_ceil_div(total, 8);
  const int grid = need < 256 * per_cu ? need : 256 * per_cu;
  *waves = grid * 8;
  hipLaunchKernelGGL((thin_long_fwd_kernel<KS, S>), dim3(grid), dim3(512), lds, stream, a, tpr, total);
  M2D_CHECK_LAUNCH("thin_long_fwd_kernel");
  return M2D_OK;
}

int m2d_thin_long_fwd(const float* x, const float* w, const float* bias, float* y, int B, int L, int ks, int stride, int pad,
                      int Lout, int act, float slope, const M2dWinView* wv, double* stats, void* ws, size_t ws_bytes,
                      hipStream_t stream) {
  ThinArgs a;
  memset(&a, 0, sizeof(a));
  if (wv) { a.xT = wv->T; a.xS = wv->S; a.xhop = wv->hop; }
  const int tpr = Lout / 32;
  if ((long long)B * tpr >= (1LL << 30) || (long long)Lout * 32 * 4 >= (1LL << 31) || (long long)L * 4 >= (1LL << 31))
    M2D_FAIL(M2D_ERR_RANGE, "m2d_conv1d_fwd (long single-channel kernel): too many tiles / a sample beyond 2 GiB");
  const int total = B * tpr;
  const size_t part = (size_t)THIN_LONG_MAX_WAVES * 64 * sizeof(float);
  if (stats) {
    if (!ws || ws_bytes < m2d_thin_long_stats_ws(B, Lout)) M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_conv1d_fwd (long single-channel kernel): no room for the statistics partials");
    a.stats = (float*)ws;
  }
  a.x = x; a.w = w; a.bias = bias; a.out = y;
  a.B = B; a.L = L; a.Cout = 32; a.ks = ks; a.stride = stride; a.pad = pad; a.Lout = Lout;
  a.act = act; a.slope = slope;
  M2dProfScope prof(M2D_FAM_GEMM, stream, 2.0 * B * Lout * 32.0 * ks, 0.0, "m2d_conv1d_fwd", 32, B * Lout, ks);
  int rc, waves = 0;
  if (thin_long_kind(1, 32, ks, stride) == 1) rc = thin_long_launch<250, 50>(a, tpr, total, stream, &waves);
  else rc = thin_long_launch<160, 4>(a, tpr, total, stream, &waves);
  if (rc) return rc;
  if (stats) return m2d_rowsums_reduce(a.stats, waves, 32, stats, (double*)((char*)a.stats + part), stream);
  return M2D_OK;
}

int m2d_thin_bwd_data(const float* dy, const float* w, float* dx, int B, int L, int Cout, int ks, int stride,
                      int pad, int Lout, const float* dy_mask, float dy_mask_slope, hipStream_t stream) {
  ThinArgs a;
  memset(&a, 0, sizeof(a));
  a.dy = dy; a.w = w; a.mask = dy_mask; a.out = dx;
  a.B = B; a.L = L; a.Cout = Cout; a.ks = ks; a.stride = stride; a.pad = pad; a.Lout = Lout;
  a.mask_slope = dy_mask_slope;
  const int nq = (L - 1 + pad) / stride + 1;
  M2dProfScope prof(M2D_FAM_POINTWISE, stream, 2.0 * B * Lout * (double)Cout * ks,
                    4.0 * B * ((double)L + (double)Cout * Lout * (dy_mask ? 2 : 1)), "thin_conv_bwd_data", 1, B * L, Cout * ks);
  a.chunk = thin_bw_chunk(B, nq);
  const dim3 grid(m2d_ceil_div(nq, a.chunk), B);
  if (dy_mask) hipLaunchKernelGGL((thin_bwd_data_mfma_kernel<25, 4, true>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((thin_bwd_data_mfma_kernel<25, 4, false>), grid, dim3(256), 0, stream, a);
  M2D_CHECK_LAUNCH("thin_bwd_data_mfma_kernel");
  return M2D_OK;
}

int m2d_thin_bwd_weight(const float* x, const float* dy, float* dw, float* dbias, int B, int L, int Cout, int ks,
                        int stride, int pad, int Lout, const float* dy_mask, float dy_mask_slope, void* ws,
                        size_t ws_bytes, const M2dWinView* wv, hipStream_t stream) {
  if (!ws || ws_bytes < m2d_thin_bwd_weight_ws(B, Cout, ks, Lout))
    M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_conv1d_bwd_weight (thin): workspace too small");
  ThinArgs a;
  memset(&a, 0, sizeof(a));
  if (wv) { a.xT = wv->T; a.xS = wv->S; a.xhop = wv->hop; }
  a.x = x; a.dy = dy; a.mask = dy_mask; a.out = (float*)ws;
  a.B = B; a.L = L; a.Cout = Cout; a.ks = ks; a.stride = stride; a.pad = pad; a.Lout = Lout;
  a.mask_slope = dy_mask_slope;
  a.chunk = thin_bw_chunk(B, Lout);
  const int nchunk = m2d_ceil_div(Lout, a.chunk);
  M2dProfScope prof(M2D_FAM_POINTWISE, stream, 2.0 * B * Lout * (double)Cout * ks,
                    4.0 * B * ((double)L + (double)Cout * Lout * (dy_mask ? 2 : 1)), "thin_conv_bwd_weight", Cout, ks, B * Lout);
  // bias partials (row 31 of the MFMA tile = ones) sit behind the weight partials in the workspace
  const size_t nblk = (size_t)nchunk * B;
  a.out2 = dbias ? (float*)ws + nblk * Cout * ks : nullptr;
  if (dy_mask)
    hipLaunchKernelGGL((thin_bwd_weight_mfma_kernel<25, 4, true>), dim3(nchunk, B), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL((thin_bwd_weight_mfma_kernel<25, 4, false>), dim3(nchunk, B), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(thin_sum_partials_kernel, dim3(m2d_ceil_div(Cout * ks, 8) + (dbias ? m2d_ceil_div(Cout, 8) : 0)),
                     dim3(256), 0, stream, (const float*)ws, dw, nchunk * B, Cout * ks, (const float*)a.out2, dbias, Cout);
  M2D_CHECK_LAUNCH("thin_bwd_weight_mfma_kernel");
  return M2D_OK;
}
