// Separable-gather GEMM engine for gfx950 (exact-fp32 MFMA, v_mfma_f32_32x32x2_f32).
//
// One kernel computes C[M,N] = A[M,K] * B[K,N] where every operand element is fetched
// through a *separable* address map
//     addr(row, k) = R(row) + F(k),   valid iff row < nrows, k valid and (optionally)
//                                     pos_r(row) + pos_k(k) in [0, lim)
// with row -> (hi, lo) = divmod(row, rdiv). The contraction index is the pair
//     k = (hi, lo),  hi in [0, nhi),  lo in [0, kdiv),  F = hi * k_hi_stride + lo * k_lo_stride
// SHARED by both operands and walked in 16-deep chunks that never straddle a `hi` boundary
// (each hi owns ceil(kdiv / 16) chunks, the last one zero-filled past kdiv). Inside a chunk
// `hi` is constant and `lo` is a run of 16, so the address of every element is
//     (per-thread constant) + (per-chunk scalar) + (loop-invariant scalar),
// i.e. one buffer load and no vector / scalar index arithmetic per element ("uniform" chunks);
// only chunks that hold a lo tail, or whose window test varies with lo, take the general
// path. Conv layers order K as (tap, channel): hi = tap, lo = channel, so the zero-padding
// window depends on the chunk only. That form covers, with padding handled by the window test:
//   conv1d forward        (A = weights packed (Cout, k, Cin), B = implicit im2col of x)
//   conv1d backward-data  (polyphase over the stride: one dense GEMM per output phase,
//                          A = weights packed (Cin, k, Cout))
//   conv1d backward-weight(K = (sample, position), split-K)
//   linear NT / NN / TN   (plain strided matrices: nhi = 1, kdiv = K)
// which are exactly the contractions the reference executes through nn.Conv1d /
// nn.Linear (phase3/archis/default.py:64-70,117-128,201-204,298-303,326-333) and their
// first and second derivatives (losses.py:40-44).
//
// Layout in HBM is the reference's own: activations (B, C, L) row-major, weights
// (Cout, Cin, k) (+ the two packed weight images above, refreshed when the weights change).
// Tiles are staged global -> registers -> LDS (double buffered, one barrier per 16-deep
// K chunk); each of the 4 waves owns a (BM/WM) x (BN/WN) sub-tile of 32x32 MFMA accumulators.
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
  unsigned nbytes;    // extent of `base` (and `mask`) in bytes, < 2^31: the staging loads are raw
                      // buffer loads whose hardware range check returns 0 for padding / tails
  int nrows;
  int rdiv;
  float rdiv_inv;
  int r_hi_stride, r_lo_stride, r_off;
  int r_pos_mul, r_pos_off;
  int k_hi_stride, k_lo_stride;  // element offset of k = (hi, lo); k_lo_stride >= 0
  int k_pos_hi, k_pos_lo;        // window position of k = hi * k_pos_hi + lo * k_pos_lo
  int lim;                       // <= 0: no window test
  // [k_safe_lo, k_safe_hi): lo values whose window test passes for EVERY valid row and hi
  // (whole range when lim <= 0 or k_pos_lo == 0: the test is then chunk-uniform or absent)
  int k_safe_lo, k_safe_hi;
  // row-fast operands only: row (ones_row_p1 - 1) reads as 1.0 for every k instead of being gathered
  // (0 = none). Backward-weight appends such a column to its B operand: C[:, that column] = sum_k A[:, k],
  // i.e. the bias gradient comes out of the same launch (A is zero wherever k is padding).
  // ones_from_hi: the row reads as 1.0 only in chunks whose hi index (backward-weight: the sample) is >= this,
  // 0.0 before: the bias gradient then sums over the samples [ones_from_hi, B) only.
  int ones_row_p1;
  int ones_from_hi;
  // Window views (audio slicing fused into the first encoder conv, utils.py:329-353 of the reference):
  // the operand is (B*T, 1, window) windows of a padded track (B, S), window t of track b starting at
  // b*S + t*hop, never materialised. The sample index n = b*T + t is then split once more:
  //   row side (forward):  hi = n, offset = (n / rdiv2) * r_hi2_stride + (n % rdiv2) * r_hi_stride
  //   k side (backward-weight, K = (n, l)): hi = n, offset = (n / kdiv2) * k_hi2_stride + (n % kdiv2) * k_hi_stride
  // (0 = plain single-level index).
  int rdiv2, r_hi2_stride;
  float rdiv2_inv;
  int kdiv2, k_hi2_stride;
};

struct M2dOutMap {
  float* out;
  const float* bias;      // bias_mode 1: bias[m], 2: bias[col]
  // Epilogue order: bias, activation, then the mask (value *= mask[addr] > 0 ? 1 : mask_slope) and the residual
  // (value += residual[addr]) - mask first by default (forward-type launches: out = act'(y) * conv(g) + skip),
  // residual first with `mask_last` (backward-data: dx = act'(x) * (conv^T(h) + skip gradient)).
  // With `sum_out` the launch has two outputs: out[addr] = the masked value WITHOUT the residual,
  // sum_out[addr] = value + residual[addr] (a TemporalBlock's second conv: relu(conv) for the backward mask and
  // x + relu(conv) for the next layer, phase3/archis/default.py:207-210).
  const float* mask;
  const float* residual;
  float* sum_out;
  int mask_last;
  // > 0: `mask` holds only the first mask_wrap elements of the output's index space and repeats behind them - element
  // `addr` reads mask[addr - mask_wrap] once addr >= mask_wrap (ONE wrap: the output spans at most 2 x mask_wrap).
  // A (2B, C, L) gradient whose two halves pass through the SAME activation masks (the audio branch of the critic:
  // the penalty's first backward and the score backward, phase3/archis/default.py:312-319 under losses.py:40-44) is
  // then one launch over 2B rows instead of two over B.
  unsigned mask_wrap;
  // set by the launcher when the tile can leave as 16-byte rows (m2d_tile_epilogue, WIDE): unit column stride, every
  // pitch / offset / column count a multiple of 4, 16-byte aligned pointers, no window / redirect column
  int wide;
  // set by the caller (sub-pixel backward-data at stride 4): rows 4 c .. 4 c + 3 of one column are four consecutive
  // addresses, which is exactly what one lane holds in four accumulator registers - the tile leaves as (4-byte
  // aligned) 16-byte stores (m2d_tile_epilogue_quad). Requires m_div == 4, m_lo_stride == 1, m_pos_mul == 1, a window
  // (c_lim > 0), M a multiple of 4, no split-K / statistics / redirect / second output / per-column bias.
  int quad;
  float mask_slope;
  float slope;            // LeakyReLU slope for act == 2
  int bias_mode;
  int act;                // 0 none, 1 ReLU, 2 LeakyReLU
  int m_stride;
  // optional two-level row map (m_div > 0): row -> (mhi, mlo) = divmod(row, m_div), row offset = mhi * m_stride +
  // mlo * m_lo_stride, and the column window test uses pos(col) + mlo * m_pos_mul. The sub-pixel form of a strided
  // backward-data conv writes row (ci, phase r) to ci * L + r and tests 4 q + r - pad against [0, L).
  int m_div, m_lo_stride, m_pos_mul;
  int cdiv;
  float cdiv_inv;
  int c_hi_stride, c_lo_stride, c_off;
  int c_pos_mul, c_pos_off, c_lim;  // c_lim <= 0: no window test on the output column
  // column (redirect_col_p1 - 1) is written raw to col_out[row] instead of through the map (0 = none)
  float* col_out;
  int redirect_col_p1;
  // optional per-row statistics of the stored values. Every wave writes, for each row of its
  // sub-tile, (sum, sum of squares) over its columns to row_part[((n_tile * WN + wn) * M + row) * 2]
  // (WN = wave columns of the tile: 2 for BM >= 64, 4 for BM = 32; plain stores - atomics onto a few
  // hundred addresses from thousands of tiles serialise: measured 5 ms per step), and
  // m2d_rowsums_reduce sums the partials per row in a fixed order into fp64 (sum, sum of squares):
  // a conv that feeds a BatchNorm hands it the batch statistics instead of a second pass over the
  // activation. Not available under split-K (the launcher then refuses to split).
  float* row_part;
  double* row_sums;  // [2 * M]: written by the launcher's reduction over row_part
};
// sums[2*row], sums[2*row + 1] (fp64) = fixed-order sums over the P partials part[p][row][0..1]
// `scratch` (optional, M2D_ROWSUMS_GROUPS * M fp64 pairs): enables the two-stage sum when P * M is large
#define M2D_ROWSUMS_GROUPS 256
#define M2D_ROWSUMS_TWO_STAGE (64LL * 1024)
int m2d_rowsums_reduce(const float* part, int P, int M, double* sums, double* scratch, hipStream_t stream);
static inline size_t m2d_rowstats_part_bytes(int M, int N) { return (size_t)((N + 127) / 128) * 4 * (size_t)M * 2 * sizeof(float); }
// partials + the two-stage sum's scratch behind them (16-byte aligned)
static inline size_t m2d_rowstats_bytes(int M, int N) {
  return ((m2d_rowstats_part_bytes(M, N) + 15) & ~(size_t)15) + (size_t)M2D_ROWSUMS_GROUPS * M * 2 * sizeof(double);
}
static inline double* m2d_rowstats_scratch(float* row_part, int M, int N) {
  return (double*)((char*)row_part + ((m2d_rowstats_part_bytes(M, N) + 15) & ~(size_t)15));
}

struct M2dGemmParams {
  M2dOperand A, B;
  M2dOutMap O;
  int M, N, K;          // K = nhi * kdiv (reported work); the kernel walks nhi * ceil(kdiv/16) chunks
  int nhi, kdiv;
  int lo_outer;        // chunk order: 0 = (hi, lo block), 1 = (lo block, hi)
  float small_tile_penalty;  // launch-plan hint (see m2d_gemm_plan); 0 = none
  int plan_kind;             // launch family for the plan's cost model: M2D_PLAN_* (0 = the general model)
  int tall_last_rb;          // > 0: phase-major sub-pixel backward-data (m2d_gemm_dl_tall_kernel): rows 32 r + ci, the last
                             // tap slot exists for the first tall_last_rb phases only (1 for k25 / stride 4)
  // conv backward-data mode (bwd_data != 0): the kernel derives, per output phase
  // r = blockIdx.z of the stride-`phases` lattice, the tap count, the K extent and the
  // q-range [qmin, qmax] of output positions j = phases*q + r - ph_pad inside [0, ph_L).
  int bwd_data;
  int phases;
  int ph_ks, ph_cout, ph_pad, ph_L, ph_batch;
  int ph_a_step;        // element offset of one tap in the A operand (0: ph_cout, the (Cin, ks, Cout) image)
  int splits;           // > 1: split-K, partial tiles go to slab[split][M*N]
  float* slab;
  // split-K without the second launch (m2d_splitk_fixup): != NULL: one arrival counter per output tile, zero on entry
  // and left zero (the stream's registered scratch, m2d_stream_scratch_set); partial tiles then sit in the slab as
  // [tile][split][register image] and the workgroup that arrives last sums them in split order and runs the epilogue
  unsigned* tickets;
  // tap-vectorised stride-4 forward conv (m2d_conv_k4_kernel): K is walked in groups of 4 consecutive taps of one
  // channel, k4_ng groups per channel; B.k_hi_stride = the channel pitch (L)
  int k4_ng;
  int k4_pair, k4_shift, k4_nlast;  // phantom-paired K order: leading phantom slots of group 0, real slots of the last group
  // > 0: the algorithmic FLOPs of the launch when the walked K holds structural zeros (phantom taps): what the
  // profiler reports instead of 2 M N K
  double work_flops;
  // workgroup -> tile map, set by the launcher (m2d_tile_of in gemm_engine.hip): 0 = tile id = workgroup id (N fastest),
  // 1 = XCD-aware grouped order (M2D_TILE_MAP=0 restores 0)
  int tile_map;
  // set by the launcher: row statistics of one-element-per-lane tiles from a second, 16-byte read of the epilogue image
  // instead of a 32-lane sum per element (M2D_STATS_NARROW_FAST=0: the round-5 form)
  int stats_narrow_fast;
};

struct M2dGemmPlan {
  int bm;      // 32, 64 or 128
  int splits;  // >= 1
  size_t ws_bytes;
};

static inline int m2d_chunks(int nhi, int kdiv) { return nhi * ((kdiv + M2D_BK - 1) / M2D_BK); }
#define M2D_PLAN_GENERAL 0
#define M2D_PLAN_BWD_WEIGHT 1  // K-streaming weight gradient: dy K-fast, K = (sample, position)
#define M2D_PLAN_BWD_DATA 2    // strided backward-data (polyphase or sub-pixel form)
M2dGemmPlan m2d_gemm_plan(int M, int N, int nchunks, int phases, bool allow_split, double small_tile_penalty = 1.0,
                          int kind = M2D_PLAN_GENERAL);
int m2d_gemm_launch(M2dGemmParams& p, bool a_kfast, bool b_kfast, bool allow_split, void* ws,
                    size_t ws_bytes, hipStream_t stream, const char* what);
// whether the LDS-direct kernels are in use (M2D_DL=0 turns them off: A/B lever)
bool m2d_dl_enabled();
// the same for the tap-vectorised stride-4 forward conv (p.k4_ng > 0; operands as documented at the kernel)
int m2d_conv_k4_launch(M2dGemmParams& p, bool allow_split, void* ws, size_t ws_bytes, hipStream_t stream,
                       const char* what);

// bytes of split-K slab a plan (bm, splits) may use: whole tiles (the in-kernel fix-up stores register images)
static inline size_t m2d_slab_bytes(int M, int N, int bm, int splits) {
  if (splits <= 1) return 0;
  return (size_t)splits * ((size_t)((M + bm - 1) / bm) * bm) * ((size_t)((N + 127) / 128) * 128) * sizeof(float);
}
// the calling stream's zero-kept scratch (tickets), or NULL / 0
unsigned* m2d_stream_scratch_get(hipStream_t stream, size_t* bytes);

static inline unsigned m2d_extent_bytes(long long elements) {
  const long long b = elements * 4;
  return b >= 0x80000000LL ? 0u : (unsigned)b;  // 0 = too large for buffer addressing (launch refuses)
}

// plain strided matrix: element (row, k) at row * row_stride + k * k_stride (use with nhi = 1, kdiv = K)
static inline void m2d_operand_plain(M2dOperand& o, const float* base, int nrows, int row_stride,
                                     int k_stride, long long elements) {
  memset(&o, 0, sizeof(o));
  o.base = base;
  o.nbytes = m2d_extent_bytes(elements);
  o.nrows = nrows;
  o.rdiv = 1;
  o.rdiv_inv = 1.f;
  o.r_hi_stride = row_stride;
  o.k_lo_stride = k_stride;
  o.k_safe_lo = 0;
  o.k_safe_hi = 0x7fffffff;
}

static inline void m2d_outmap_plain(M2dOutMap& o, float* out, int m_stride, int c_stride) {
  memset(&o, 0, sizeof(o));
  o.out = out;
  o.m_stride = m_stride;
  o.cdiv = 1;
  o.cdiv_inv = 1.f;
  o.c_hi_stride = c_stride;
}
