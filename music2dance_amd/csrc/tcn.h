// Dedicated kernels for the pose critic's TemporalBlock convolutions (Conv1d(C, 128, k, stride 1, "same" padding),
// phase3/archis/default.py:195-210 and phase2/archis/default.py:27-49 of the reference) and their two backward halves.
//
// Why a kernel family of their own (round 6): as GEMMs these launches are M = 128 x N = (rows * T) x K = 7 * 128 -
// 5 GFLOP at 3B rows, 1.8 at B rows - and the general engine tiles them into 180-360 workgroups of 14-56 chunks whose
// per-workgroup prologue, split-K fix-up and two-stage staging round trip (1.46 us per chunk against 0.85 us of MFMA
// for a lone workgroup) cost as much as the arithmetic: 49-73 TFLOP/s for three rounds. What the
// general engine cannot use is the structure of a stride-1 convolution:
//   * forward / backward-data / tangent (m2d_tcn_conv_kernel): the B operand of all k taps is ONE activation tile with a
//     halo. It is staged ONCE per workgroup (128 channels x (NT + halo) positions, 22-55 KB of LDS) and every tap's
//     fragments are read from it at an offset of one float - the general engine stages the same lines k times. Only the
//     weights stream (8 KB per 16-deep chunk, one 16-byte LDS-DMA instruction per wave, four stages in flight with
//     counted vmcnt). One workgroup = all 128 output channels x NT positions x the WHOLE K: no split-K, no slab, no
//     fix-up, no tickets. NT in {96, 64, 32} is chosen so that the launch is one round of <= 256 workgroups
//     (3B * 120 = 23 040 = 240 x 96). Eight waves = 4 row blocks x 2 K halves: two waves per SIMD hide each other's
//     LDS latency, the halves meet once, in LDS, after the loop.
//   * weight gradient (m2d_tcn_wgrad_kernel): K = positions. A workgroup owns 16 input channels (x all k taps = 112
//     columns for k7) and a run of whole samples; per sample it stages dy (128 x L) and its 16 x (L + halo) slice of x
//     once and reads the k shifted B fragments from that one image. Partial tiles of the sample runs are summed in a
//     fixed order by a second, chip-wide pass (deterministic).
#pragma once
#include "m2d_common.h"

struct M2dTcnConv {
  const float* x;        // (B, Cin, L)
  const float* x_mask;   // optional, shape of x: x is read as x * (mask > 0 ? 1 : x_mask_slope)
  const float* wimg;     // K-major weight image [(c * ks + tap) * 128 + m], c < Cin (m2d_conv1d_pack_weights)
  const float* bias;     // optional, 128
  float* out;            // (B, 128, L)
  const float* out_mask; // optional, shape of out
  const float* residual; // optional, shape of out
  float* sum_out;        // optional second output (needs residual): out = masked value, sum_out = value + residual
  float x_mask_slope, out_mask_slope, slope;
  int act;               // 0 none, 1 ReLU, 2 LeakyReLU(slope)
  int mask_last;         // 0: out = mask * act(conv + bias) + residual; 1: out = mask * (conv + residual)
  int tap_rev;           // 1: tap t reads image tap ks - 1 - t (backward-data over the (Cout, ks, Cin) image)
  unsigned out_mask_wrap; // > 0: out_mask holds only the first out_mask_wrap elements and repeats behind them (ONE wrap:
                         // m2d_conv1d_bwd_data_shared_mask, the two halves of a batch behind the same activation masks)
  int B, Cin, L;
};

// -> 0 (not applicable: the caller takes the general engine) or the tile width chosen
int m2d_tcn_conv_tile(int B, int Cin, int L, int Cout, int ks, int stride, int pad);
int m2d_tcn_conv_launch(const M2dTcnConv& p, int ks, int nt, hipStream_t stream, const char* what);

struct M2dTcnWgrad {
  const float* x;        // (B, Cin, L)
  const float* dy;       // (B, 128, L)
  const float* dy_mask;  // optional, shape of dy
  float dy_mask_slope;
  float* dw;             // (128, Cin, ks)
  float* dbias;          // optional, 128: sum of (masked) dy over the samples [bias_from, B)
  int bias_from;
  int B, Cin, L;
};
int m2d_tcn_wgrad_applicable(int B, int Cin, int L, int Cout, int ks, int stride, int pad);
size_t m2d_tcn_wgrad_ws_bytes(int B, int Cin, int L, int ks);
int m2d_tcn_wgrad_launch(const M2dTcnWgrad& p, int ks, void* ws, size_t ws_bytes, hipStream_t stream, const char* what);
