// Separable-gather GEMM engine for gfx950 (exact-fp32 MFMA, v_mfma_f32_32x32x2_f32).
//
// One kernel computes C[M,N] = A[M,K] * B[K,N] where every operand element is fetched
// through a *separable* address map
//     addr(row, k) = R(row) + F(k),   valid iff row < nrows, k < K and (optionally)
//                                     pos_r(row) + pos_k(k) in [0, lim)
// with row -> (hi, lo) = divmod(row, rdiv) and k -> (hi, lo) = divmod(k, kdiv).
// That single form covers, with zero-padding handled by the window test:
//   conv1d forward        (A = weights, B = implicit im2col of x)
//   conv1d backward-data  (polyphase over the stride: one dense GEMM per output phase)
//   conv1d backward-weight(split-K over batch*length)
//   linear NT / NN / TN   (plain strided matrices)
// which are exactly the contractions the reference executes through nn.Conv1d /
// nn.Linear (phase3/archis/default.py:64-70,117-128,201-204,298-303,326-333) and their
// first and second derivatives (losses.py:40-44).
//
// Layout in HBM is the reference's own: activations (B, C, L) row-major, weights
// (Cout, Cin, k). Tiles are staged global -> registers -> LDS (double buffered, one
// barrier per 16-deep K chunk); each of the 4 waves owns a (BM/WM) x (BN/WN) sub-tile
// made of 32x32 MFMA accumulators.
#pragma once
#include "m2d_common.h"

#ifndef M2D_BK
#define M2D_BK 16
#endif
#define M2D_LDPAD (M2D_BK == 16 ? 2 : 1)

struct M2dOperand {
  const float* base;
  const float* mask;  // optional, same addressing: value *= (mask > 0 ? 1 : mask_slope)
  float mask_slope;
  unsigned nbytes;    // extent of `base` (and `mask`) in bytes, <= 0xFFFFFFF0: the staging loads are
                      // raw buffer loads whose hardware range check returns 0 for padding / tails
  int nrows;
  int rdiv;
  float rdiv_inv;
  int r_hi_stride, r_lo_stride, r_off;
  int r_pos_mul, r_pos_off;
  int kdiv;
  float kdiv_inv;
  int k_hi_stride, k_lo_stride;
  int k_pos_mul;
  int lim;  // <= 0: no window test
};

struct M2dOutMap {
  float* out;
  const float* bias;      // bias_mode 1: bias[m], 2: bias[col]
  const float* mask;      // optional: value *= (mask[addr] > 0 ? 1 : mask_slope), applied last
  const float* residual;  // optional: value += residual[addr] (after activation)
  float mask_slope;
  float slope;            // LeakyReLU slope for act == 2
  int bias_mode;
  int act;                // 0 none, 1 ReLU, 2 LeakyReLU
  int m_stride;
  int cdiv;
  float cdiv_inv;
  int c_hi_stride, c_lo_stride, c_off;
  int c_pos_mul, c_pos_off, c_lim;  // c_lim <= 0: no window test on the output column
};

struct M2dGemmParams {
  M2dOperand A, B;
  M2dOutMap O;
  int M, N, K;
  // conv backward-data mode (bwd_data != 0): the kernel derives, per output phase
  // r = blockIdx.z of the stride-`phases` lattice, the tap count, the K extent and the
  // q-range [qmin, qmax] of output positions j = phases*q + r - ph_pad inside [0, ph_L).
  int bwd_data;
  int phases;
  int ph_ks, ph_cout, ph_pad, ph_L, ph_batch;
  int splits;           // > 1: split-K, partial tiles go to slab[split][M*N]
  float* slab;
};

struct M2dGemmPlan {
  int bm;      // 32, 64 or 128
  int splits;  // >= 1
  size_t ws_bytes;
};

M2dGemmPlan m2d_gemm_plan(int M, int N, int K, int phases, bool allow_split);
int m2d_gemm_launch(M2dGemmParams& p, bool a_kfast, bool b_kfast, bool allow_split, void* ws,
                    size_t ws_bytes, hipStream_t stream, const char* what);

static inline unsigned m2d_extent_bytes(long long elements) {
  const long long b = elements * 4;
  return b > 0xFFFFFFF0LL ? 0u : (unsigned)b;  // 0 = too large for buffer addressing (launch refuses)
}

static inline void m2d_operand_plain(M2dOperand& o, const float* base, int nrows, int row_stride,
                                     int k_stride, long long elements) {
  memset(&o, 0, sizeof(o));
  o.base = base;
  o.nbytes = m2d_extent_bytes(elements);
  o.nrows = nrows;
  o.rdiv = 1;
  o.rdiv_inv = 1.f;
  o.r_hi_stride = row_stride;
  o.kdiv = 1;
  o.kdiv_inv = 1.f;
  o.k_hi_stride = k_stride;
}

static inline void m2d_outmap_plain(M2dOutMap& o, float* out, int m_stride, int c_stride) {
  memset(&o, 0, sizeof(o));
  o.out = out;
  o.m_stride = m_stride;
  o.cdiv = 1;
  o.cdiv_inv = 1.f;
  o.c_hi_stride = c_stride;
}
