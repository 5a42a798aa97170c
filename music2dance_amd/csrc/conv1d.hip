// C-ABI entry points that express conv1d forward / backward-data / backward-weight and
// the three linear GEMMs as separable-gather GEMMs (gemm_engine.h).
//
// Semantics follow torch.nn.Conv1d / nn.Linear as the reference uses them
// (cross-correlation, zero padding, dilation 1, groups 1):
//   y[n,co,l]  = b[co] + sum_{ci,kk} W[co,ci,kk] * x[n,ci,l*s - p + kk]
//   dx[n,ci,j] = sum_{co,kk : j = l*s - p + kk} W[co,ci,kk] * dy[n,co,l]
//   dW[co,ci,kk] = sum_{n,l} dy[n,co,l] * x[n,ci,l*s - p + kk]
// The three are closed under differentiation, which is what the gradient penalty's
// double backward needs (losses.py:40-44): every optional `*_mask` argument multiplies
// an operand / the result by (mask > 0 ? 1 : slope), i.e. by the derivative of the fused
// ReLU / LeakyReLU, so the first- and second-order chains need no extra HBM pass.
#include "gemm_engine.h"
#include "tcn.h"

static inline int conv_out_len(int L, int ks, int stride, int pad) {
  const int span = L + 2 * pad - ks;
  return span < 0 ? 0 : span / stride + 1;
}

static inline bool fits_i32(long long v) { return v >= 0 && v < 2147483647LL; }

// conv1d_thin.hip: vector-ALU kernels for Cin = 1, k = 25, stride 4 (HBM-bound layers)
// Window view of a padded track (B, S): the logical input is (B*T, 1, window), window t of track b
// starting at b*S + t*hop (utils.slice_audio_batch of the reference, never materialised). T == 0: dense.
struct M2dWinView {
  int T, S, hop;
};
bool m2d_thin_applicable(int Cin, int Cout, int ks, int stride);
bool m2d_thin_long_applicable(int Cin, int Cout, int ks, int stride, int pad, int L);
size_t m2d_thin_long_stats_ws(int B, int Lout);
int m2d_thin_long_fwd(const float* x, const float* w, const float* bias, float* y, int B, int L, int ks, int stride, int pad,
                      int Lout, int act, float slope, const M2dWinView* wv, double* stats, void* ws, size_t ws_bytes,
                      hipStream_t stream);
size_t m2d_thin_bwd_weight_ws(int B, int Cout, int ks, int Lout);
size_t m2d_thin_fwd_stats_ws(int B, int Lout);
int m2d_thin_fwd(const float* x, const float* w, const float* bias, float* y, int B, int L, int Cout, int ks,
                 int stride, int pad, int Lout, int act, float slope, const float* out_mask, float out_mask_slope,
                 const M2dWinView* wv, double* stats, void* ws, size_t ws_bytes, hipStream_t stream);
int m2d_thin_bwd_data(const float* dy, const float* w, float* dx, int B, int L, int Cout, int ks, int stride,
                      int pad, int Lout, const float* dy_mask, float dy_mask_slope, hipStream_t stream);
int m2d_thin_bwd_weight(const float* x, const float* dy, float* dw, float* dbias, int B, int L, int Cout, int ks,
                        int stride, int pad, int Lout, const float* dy_mask, float dy_mask_slope, void* ws,
                        size_t ws_bytes, const M2dWinView* wv, hipStream_t stream);

// Forward conv as a GEMM. Cin >= 16: K ordered (tap, channel) - hi = tap, lo = ci - over the
// packed weights wp (Cout, ks, Cin): every 16-chunk sits on one tap, so the padding window is
// chunk-uniform. Narrow inputs (the encoders' first layer, Cin = 1, k = 250) keep the
// natural order (channel, tap) over w itself.
static inline bool conv_uses_packed(int Cin) { return Cin >= M2D_BK; }
static inline float fwd_tile_penalty(int stride) { return stride > 1 ? 1.25f : 1.f; }

static void fill_fwd(M2dGemmParams& p, const float* x, const float* w, const float* wp, float* y, int B, int Cin,
                     int L, int Cout, int ks, int stride, int pad, int Lout, const M2dWinView* wv = nullptr) {
  memset(&p, 0, sizeof(p));
  p.M = Cout;
  p.N = B * Lout;
  p.K = Cin * ks;
  p.phases = 1;
  const bool packed = conv_uses_packed(Cin);
  M2dOperand& b = p.B;
  b.base = x;
  b.nbytes = m2d_extent_bytes((long long)B * Cin * L);
  b.nrows = p.N;
  b.rdiv = Lout;
  b.rdiv_inv = 1.f / (float)Lout;
  b.r_hi_stride = Cin * L;
  b.r_lo_stride = stride;
  b.r_off = -pad;
  b.r_pos_mul = stride;
  b.r_pos_off = -pad;
  b.lim = L;
  if (wv && wv->T > 0) {  // Cin == 1: sample n = (track, window t) at track * S + t * hop
    b.nbytes = m2d_extent_bytes((long long)(B / wv->T) * wv->S);
    b.r_hi_stride = wv->hop;
    b.rdiv2 = wv->T;
    b.rdiv2_inv = 1.f / (float)wv->T;
    b.r_hi2_stride = wv->S;
  }
  if (packed) {
    p.nhi = ks;
    p.kdiv = Cin;
    p.lo_outer = 1;
    p.small_tile_penalty = fwd_tile_penalty(stride);
    // K-major weight image (Cin, ks, Cout): A(m = co, k = (kk,ci)) = wp[ci*ks*Cout + kk*Cout + co] - rows contiguous,
    // so both operands are row-fast and the launch can stage straight into the LDS (m2d_gemm_dl_kernel)
    m2d_operand_plain(p.A, wp, Cout, 1, ks * Cout, (long long)Cout * Cin * ks);
    p.A.k_hi_stride = Cout;
    // B(k = (kk,ci), col = (n,l)) = x[n*Cin*L + ci*L + l*s - p + kk]
    b.k_hi_stride = 1;
    b.k_lo_stride = L;
    b.k_pos_hi = 1;
    b.k_pos_lo = 0;
    b.k_safe_lo = 0;
    b.k_safe_hi = 0x7fffffff;
  } else {
    p.nhi = Cin;
    p.kdiv = ks;
    // A(m = co, k = (ci,kk)) = w[co*Cin*ks + ci*ks + kk]
    m2d_operand_plain(p.A, w, Cout, Cin * ks, 1, (long long)Cout * Cin * ks);
    p.A.k_hi_stride = ks;
    b.k_hi_stride = L;
    b.k_lo_stride = 1;
    b.k_pos_hi = 0;
    b.k_pos_lo = 1;
    // taps kk whose position l*s - pad + kk lies in [0, L) for every l
    b.k_safe_lo = pad;
    b.k_safe_hi = L - (Lout - 1) * stride + pad;
  }
  // out(m = co, col = (n,l)) = y[n*Cout*Lout + co*Lout + l]
  m2d_outmap_plain(p.O, y, Lout, 1);
  p.O.cdiv = Lout;
  p.O.cdiv_inv = 1.f / (float)Lout;
  p.O.c_hi_stride = Cout * Lout;
  p.O.c_lo_stride = 1;
}

// wf[co][kk][ci] = wb[ci][kk][co] = w[co][ci][kk]
__global__ void __launch_bounds__(256) m2d_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wf,
                                                                 float* __restrict__ wb, int Cout, int Cin, int ks) {
  const size_t total = (size_t)Cout * Cin * ks;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    // idx enumerates the forward image (co, kk, ci): coalesced writes of wf, strided reads of w
    const int ci = (int)(idx % Cin);
    const size_t t = idx / Cin;
    const int kk = (int)(t % ks);
    const int co = (int)(t / ks);
    const float v = w[((size_t)co * Cin + ci) * ks + kk];
    if (wf) wf[idx] = v;
    if (wb) wb[((size_t)ci * ks + kk) * Cout + co] = v;
  }
}

// Sub-pixel image of a strided backward-data conv: wsp[(t * Cout + co) * (Cin * s) + ci * s + r] = w[co][ci][r + s * t]
// (zero for taps >= ks): K-major, rows (ci, phase r) contiguous; phase_major: row = r * Cin + ci (m2d_gemm_dl_tall_kernel)
__global__ void __launch_bounds__(256) m2d_pack_weights_subpixel_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                                          int Cout, int Cin, int ks, int s, int nt, int phase_major) {
  const size_t total = (size_t)nt * Cout * Cin * s;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int m = (int)(idx % (size_t)(Cin * s));
    const size_t k = idx / (size_t)(Cin * s);
    // phase-major: tiles of 128 rows = 32 channels x s phases, row = 128 tile + 32 r + (ci % 32)
    const int ci = phase_major ? (m / (32 * s)) * 32 + m % 32 : m / s, r = phase_major ? (m % (32 * s)) / 32 : m - ci * s;
    const int co = (int)(k % Cout), t = (int)(k / Cout);
    const int tap = r + s * t;
    out[idx] = tap < ks ? w[((size_t)co * Cin + ci) * ks + tap] : 0.f;
  }
}

// Strided backward-data in its sub-pixel form when the layer has few input channels: rows (ci, phase), one GEMM for all
// phases (M = Cin * s instead of s GEMMs with M = Cin). Measured need: the 32-channel audio layer ran 32-row tiles at
// 67 TFLOP/s, the worst large launch of the step. Cost: ceil(ks / s) * s tap slots instead of ks (k25 / s4: 28, + 12 %).
static inline bool subpixel_tall(int Cin, int Cout, int ks, int stride);
static inline bool bwd_subpixel(int Cin, int Cout, int ks, int stride, bool masked_dy = false) {
  static const bool on = [] { const char* e = getenv("M2D_SUBPIXEL"); return !(e && e[0] == '0'); }();
  static const int max_rows = [] { const char* e = getenv("M2D_SUBPIXEL_MAXROWS"); return e ? atoi(e) : 128; }();  // A/B lever
  static const int tall_max_cin = [] { const char* e = getenv("M2D_SUBPIXEL_TALL_MAXCIN"); return e ? atoi(e) : 64; }();  // A/B lever (measured: 64 -> 128 layer 658 -> 587 us, the 128- and 256-channel layers lose 7 - 12 %)
  if (on && !masked_dy && Cin > 32 && Cin <= tall_max_cin && subpixel_tall(Cin, Cout, ks, stride)) return true;
  return on && stride > 1 && Cin * stride <= max_rows && Cin * stride >= 64 && Cout >= 16 && ks > stride;
}
// ... and without the phantom taps of the last slot where the tile is exactly the four phases of 32 channels (the audio
// critic's 32 -> 64 layer, the WaveGAN encoder's): M2D_SUBPIXEL_TALL=0 keeps the (ci, r) order
static inline bool subpixel_tall(int Cin, int Cout, int ks, int stride) {
  static const bool on = [] { const char* e = getenv("M2D_SUBPIXEL_TALL"); return !(e && e[0] == '0'); }();
  const int nt = (ks + stride - 1) / stride;
  // (the phase-major image is readable by m2d_gemm_dl_tall_kernel only: with M2D_DL=0 the (ci, r) order stays - ADVICE r5)
  return on && m2d_dl_enabled() && Cin % 32 == 0 && stride == 4 && ks - stride * (nt - 1) == 1 && nt >= 2;
}
static inline size_t subpixel_bytes(int Cout, int Cin, int ks, int stride) {
  return ((((size_t)((ks + stride - 1) / stride) * Cout * Cin * stride) * sizeof(float)) + 255) & ~(size_t)255;
}

static int pack_weights(const float* w, float* wf, float* wb, int Cout, int Cin, int ks, hipStream_t stream) {
  const size_t total = (size_t)Cout * Cin * ks;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(m2d_pack_weights_kernel, dim3(blocks), dim3(256), 0, stream, w, wf, wb, Cout, Cin, ks);
  M2D_CHECK_LAUNCH("m2d_pack_weights_kernel");
  return M2D_OK;
}

static inline size_t pack_bytes(int Cout, int Cin, int ks) {
  return (((size_t)Cout * Cin * ks * sizeof(float)) + 255) & ~(size_t)255;
}

extern "C" {

// Packed weight images the forward (wf: (Cout, ks, Cin)) and backward-data (wb: (Cin, ks, Cout))
// GEMMs read; either pointer may be NULL. Callers that keep them across calls (and refresh
// them when the weights change) pass them as `w_packed`; otherwise each call packs into its
// workspace.
int m2d_conv1d_pack_weights(const float* w, float* w_fwd, float* w_bwd, int Cout, int Cin, int ks, void* stream) {
  if (Cout <= 0 || Cin <= 0 || ks <= 0 || !w) M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_pack_weights: bad arguments");
  if (!fits_i32((long long)Cout * Cin * ks)) M2D_FAIL(M2D_ERR_RANGE, "m2d_conv1d_pack_weights: too large");
  if (!w_fwd && !w_bwd) return M2D_OK;
  // the forward reads the K-major image (Cin, ks, Cout), backward-data (Cout, ks, Cin): see fill_fwd / m2d_conv1d_bwd_data
  return pack_weights(w, /*(Cout, ks, Cin) ->*/ w_bwd, /*(Cin, ks, Cout) ->*/ w_fwd, Cout, Cin, ks, (hipStream_t)stream);
}

// Replaces nn.Conv1d forward (+ fused bias / ReLU / LeakyReLU / residual add):
// phase3/archis/default.py:78-82,137-143,207-210,312-319,342-346.
// `w_packed` (optional): the forward image (Cin, ks, Cout) of w from m2d_conv1d_pack_weights.
// `out_mask` (optional, shape of y) multiplies the result by (mask>0 ? 1 : out_mask_slope):
// it is the d/d(dy) branch of backward-data's derivative (double backward of the GP).
static int conv1d_fwd_impl(const float* x, const float* w, const float* w_packed, const float* bias, float* y, int B,
                           int Cin, int L, int Cout, int ks, int stride, int pad, int act, float slope,
                           const float* residual, const float* out_mask, float out_mask_slope, void* ws,
                           size_t ws_bytes, void* stream, const M2dWinView* wv, double* stats,
                           float* sum_out = nullptr) {
  if (sum_out && !residual) M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_fwd: sum_out needs a residual");

  if (B <= 0 || Cin <= 0 || Cout <= 0 || ks <= 0 || stride <= 0 || pad < 0 || L <= 0)
    M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_fwd: bad shape B=%d Cin=%d L=%d Cout=%d k=%d s=%d p=%d", B, Cin, L,
             Cout, ks, stride, pad);
  const int Lout = conv_out_len(L, ks, stride, pad);
  if (Lout <= 0) M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_fwd: empty output (L=%d k=%d s=%d p=%d)", L, ks, stride, pad);
  if (!fits_i32((long long)B * Cin * L) || !fits_i32((long long)B * Cout * Lout))
    M2D_FAIL(M2D_ERR_RANGE, "m2d_conv1d_fwd: tensor exceeds 2^31 elements");
  if (!residual && !sum_out && !out_mask && m2d_thin_long_applicable(Cin, Cout, ks, stride, pad, L))
    return m2d_thin_long_fwd(x, w, bias, y, B, L, ks, stride, pad, Lout, act, slope, wv, stats, ws, ws_bytes, (hipStream_t)stream);
  if (!residual && !sum_out && m2d_thin_applicable(Cin, Cout, ks, stride))
    return m2d_thin_fwd(x, w, bias, y, B, L, Cout, ks, stride, pad, Lout, act, slope, out_mask, out_mask_slope,
                        wv, stats, ws, ws_bytes, (hipStream_t)stream);
  if (wv && (Cin != 1 || (Lout == 1 && pad == 0 && L == ks)))
    M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_fwd_windows: window views are single-channel, non-degenerate convolutions");
  if (Lout == 1 && pad == 0 && L == ks) {
    // full-length kernel (fconv / l6 / last encoder conv): y[n,co] = b[co] + sum_k x[n,k] W[co,k], k = (ci,kk):
    // a plain NT GEMM over the natural layouts (both operands contiguous in k)
    M2dGemmParams p;
    memset(&p, 0, sizeof(p));
    p.M = Cout;
    p.N = B;
    p.K = Cin * ks;
    p.nhi = 1;
    p.kdiv = p.K;
    p.phases = 1;
    m2d_operand_plain(p.A, w, Cout, p.K, 1, (long long)Cout * p.K);
    m2d_operand_plain(p.B, x, B, p.K, 1, (long long)B * p.K);
    m2d_outmap_plain(p.O, y, 1, Cout);
    p.O.bias = bias;
    p.O.bias_mode = bias ? 1 : 0;
    p.O.act = act;
    p.O.slope = slope;
    p.O.residual = residual;
    p.O.sum_out = sum_out;
    p.O.mask = out_mask;
    p.O.mask_slope = out_mask_slope;
    if (stats) {   // the statistics partials first, split-K slabs (one-launch form only) behind them
      const size_t st = m2d_rowstats_bytes(p.M, p.N);
      if (!ws || ws_bytes < st) M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_conv1d_fwd: no room for the statistics partials");
      p.O.row_part = (float*)ws;
      p.O.row_sums = stats;
      ws = (char*)ws + st;
      ws_bytes -= st;
    }
    return m2d_gemm_launch(p, true, true, true, ws, ws_bytes, (hipStream_t)stream, "m2d_conv1d_fwd");
  }
  if (conv_uses_packed(Cin) && !w_packed) {
    const size_t pb = pack_bytes(Cout, Cin, ks);
    if (!ws || ws_bytes < pb) M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_conv1d_fwd: workspace too small to pack the weights");
    const int rc = pack_weights(w, nullptr, (float*)ws, Cout, Cin, ks, (hipStream_t)stream);
    if (rc) return rc;
    w_packed = (const float*)ws;
    ws = (char*)ws + pb;
    ws_bytes -= pb;
  }
  if (!wv && !stats) {
    // the pose critic's TemporalBlock convolutions (stride 1, "same" padding, 128 output channels): csrc/tcn.hip
    const int nt = m2d_tcn_conv_tile(B, Cin, L, Cout, ks, stride, pad);
    if (nt > 0) {
      M2dTcnConv t;
      memset(&t, 0, sizeof(t));
      t.x = x;
      t.wimg = w_packed;
      t.bias = bias;
      t.out = y;
      t.out_mask = out_mask;
      t.out_mask_slope = out_mask_slope;
      t.residual = residual;
      t.sum_out = sum_out;
      t.act = act;
      t.slope = slope;
      t.B = B;
      t.Cin = Cin;
      t.L = L;
      return m2d_tcn_conv_launch(t, ks, nt, (hipStream_t)stream, "m2d_conv1d_fwd");
    }
  }
  M2dGemmParams p;
  fill_fwd(p, x, w, w_packed, y, B, Cin, L, Cout, ks, stride, pad, Lout, wv);
  p.O.bias = bias;
  p.O.bias_mode = bias ? 1 : 0;
  p.O.act = act;
  p.O.slope = slope;
  p.O.residual = residual;
  p.O.sum_out = sum_out;
  p.O.mask = out_mask;
  p.O.mask_slope = out_mask_slope;
  if (stats) {   // the statistics partials first, split-K slabs (one-launch form only) behind them
    const size_t st = m2d_rowstats_bytes(p.M, p.N);
    if (!ws || ws_bytes < st) M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_conv1d_fwd: no room for the statistics partials");
    p.O.row_part = (float*)ws;
    p.O.row_sums = stats;
    ws = (char*)ws + st;
    ws_bytes -= st;
  }
  return m2d_gemm_launch(p, /*a_kfast=*/!conv_uses_packed(Cin), /*b_kfast=*/false, /*allow_split=*/true,
                         ws, ws_bytes, (hipStream_t)stream, "m2d_conv1d_fwd");
}

int m2d_conv1d_fwd(const float* x, const float* w, const float* w_packed, const float* bias, float* y, int B,
                   int Cin, int L, int Cout, int ks, int stride, int pad, int act, float slope,
                   const float* residual, const float* out_mask, float out_mask_slope, double* stats, void* ws,
                   size_t ws_bytes, void* stream) {
  return conv1d_fwd_impl(x, w, w_packed, bias, y, B, Cin, L, Cout, ks, stride, pad, act, slope, residual, out_mask,
                         out_mask_slope, ws, ws_bytes, stream, nullptr, stats);
}

// The same launch with TWO outputs (a TemporalBlock's second conv, phase3/archis/default.py:207-210: x + relu(conv)):
//   y = mask * act(conv(x) + bias)            (what the backward pass needs as its activation mask)
//   sum_out = y + residual                    (what the next layer reads)
int m2d_conv1d_fwd_sum(const float* x, const float* w, const float* w_packed, const float* bias, float* y,
                       float* sum_out, int B, int Cin, int L, int Cout, int ks, int stride, int pad, int act,
                       float slope, const float* residual, const float* out_mask, float out_mask_slope, void* ws,
                       size_t ws_bytes, void* stream) {
  if (!sum_out || !residual) M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_fwd_sum: needs sum_out and residual");
  return conv1d_fwd_impl(x, w, w_packed, bias, y, B, Cin, L, Cout, ks, stride, pad, act, slope, residual, out_mask,
                         out_mask_slope, ws, ws_bytes, stream, nullptr, nullptr, sum_out);
}

}  // extern "C"

// ---- tap-vectorised stride-4 forward (gemm_engine: m2d_conv_k4_kernel) ------------------------------------------------
static inline int k4_shift(int pad) { return ((pad + 3) / 4) * 4 - pad; }                 // P - pad: phantom taps in front
static inline int k4_groups(int ks, int pad) { return (ks + k4_shift(pad) + 3) / 4; }       // tap groups per channel

// real slots of the last tap group; and whether the layer's K is walked in the phantom-paired order (m2d_conv_k4_kernel):
// it has partial groups, at least two full groups between them (the cursor wraps at most twice per chunk), and channel
// blocks of four (M2D_K4_PAIR=0: A/B lever)
static inline int k4_nlast(int ks, int pad) { return ks + k4_shift(pad) - 4 * (k4_groups(ks, pad) - 1); }
static inline int k4_paired(int Cin, int ks, int pad) {
  static const bool on = [] { const char* e = getenv("M2D_K4_PAIR"); return !(e && e[0] == '0'); }();
  // (the kernel's partial-group loops are instantiated for slots [1, 4) and [0, 2): k25 / pad 11, every layer that takes this path)
  return on && (Cin % 4) == 0 && k4_groups(ks, pad) >= 4 && k4_shift(pad) == 1 && k4_nlast(ks, pad) == 2;
}

// Wk4[kg * Cout * 4 + co * 4 + m] = W[co][ci][4 g + m - (P - pad)], zero for taps outside [0, ks); plain order
// kg = ci * NG + g, paired order: see m2d_conv_k4_kernel (all full groups channel-major, then the g = 0 groups, then
// the g = NG - 1 groups)
__global__ void __launch_bounds__(256) m2d_pack_weights_k4_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                                    int Cout, int Cin, int ks, int ng, int shift, int pair) {
  const size_t total = (size_t)Cin * ng * Cout * 4;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int m = (int)(idx & 3);
    const size_t r = idx >> 2;
    const int co = (int)(r % Cout);
    const size_t kg = r / Cout;
    int g = (int)(kg % ng), ci = (int)(kg / ng);
    if (pair) {  // regions: Cin * (ng - 2) full groups channel-major, then every channel's g = 0, then every g = ng - 1
      const int nfull = ng - 2;
      const size_t r1 = (size_t)Cin * nfull;
      if (kg < r1) {
        ci = (int)(kg / nfull);
        g = 1 + (int)(kg - (size_t)ci * nfull);
      } else {
        ci = (int)((kg - r1) % Cin);
        g = kg - r1 < (size_t)Cin ? 0 : ng - 1;
      }
    }
    const int tap = 4 * g + m - shift;
    out[idx] = (tap >= 0 && tap < ks) ? w[((size_t)co * Cin + ci) * ks + tap] : 0.f;
  }
}

extern "C" {

// 1 when m2d_conv1d_fwd_k4 takes the layer: stride 4, L a multiple of 4 (tap groups never straddle a row end), Cin a
// multiple of 4 in [16, 64], Cout >= 64, not a full-length kernel. Cin <= 64: measured at B = 64 (one box, alternating
// runs), generic -> tap-vectorised: 32 -> 64 channels 402 -> 365 us, 64 -> 128: 365 -> 348, 128 -> 256: 325 -> 338 .. 370,
// 256 -> 512: 301 -> 317. The kernel runs 97 - 111 TFLOP/s on the 28 tap slots it multiplies, but 3 of them are phantom
// (k25: 12 %), and the wide layers' gathers were already served from L1 by the generic engine.
int m2d_conv1d_k4_applicable(int Cin, int L, int Cout, int ks, int stride, int pad) {
  const int Lout = conv_out_len(L, ks, stride, pad);
  static const int max_cin = [] { const char* e = getenv("M2D_K4_MAXCIN"); return e ? atoi(e) : 64; }();  // A/B lever
  return stride == 4 && (L % 4) == 0 && (Cin % 4) == 0 && Cin >= 16 && Cin <= max_cin && Cout >= 64 && Lout > 1 && ks >= 4 &&
         k4_groups(ks, pad) >= 2;
}

// elements of the packed image m2d_conv1d_pack_weights_k4 writes: Cin * NG * Cout * 4
size_t m2d_conv1d_k4_packed_elems(int Cout, int Cin, int ks, int pad) {
  return (size_t)Cin * k4_groups(ks, pad) * Cout * 4;
}

int m2d_conv1d_pack_weights_k4(const float* w, float* out, int Cout, int Cin, int ks, int pad, void* stream) {
  if (Cout <= 0 || Cin <= 0 || ks <= 0 || pad < 0 || !w || !out) M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_pack_weights_k4: bad arguments");
  const size_t total = m2d_conv1d_k4_packed_elems(Cout, Cin, ks, pad);
  if (!fits_i32((long long)total)) M2D_FAIL(M2D_ERR_RANGE, "m2d_conv1d_pack_weights_k4: too large");
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(m2d_pack_weights_k4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin, ks,
                     k4_groups(ks, pad), k4_shift(pad), k4_paired(Cin, ks, pad));
  M2D_CHECK_LAUNCH("m2d_pack_weights_k4_kernel");
  return M2D_OK;
}

// nn.Conv1d forward (stride 4) through the tap-vectorised kernel; `w_k4` from m2d_conv1d_pack_weights_k4 (same pad).
// Epilogue arguments as m2d_conv1d_fwd / m2d_conv1d_fwd_sum (sum_out optional). Workspace: m2d_conv1d_workspace_bytes(0, ...).
int m2d_conv1d_fwd_k4(const float* x, const float* w_k4, const float* bias, float* y, float* sum_out, int B, int Cin,
                      int L, int Cout, int ks, int stride, int pad, int act, float slope, const float* residual,
                      const float* out_mask, float out_mask_slope, double* stats, void* ws, size_t ws_bytes,
                      void* stream) {
  if (!m2d_conv1d_k4_applicable(Cin, L, Cout, ks, stride, pad))
    M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_fwd_k4: layer not eligible (stride 4, L %% 4 == 0, Cin %% 4 == 0, Cin >= 16, Cout >= 64)");
  if (B <= 0 || !x || !w_k4 || !y) M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_fwd_k4: bad arguments");
  if (sum_out && !residual) M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_fwd_k4: sum_out needs a residual");
  const int Lout = conv_out_len(L, ks, stride, pad);
  if (!fits_i32((long long)B * Cin * L) || !fits_i32((long long)B * Cout * Lout))
    M2D_FAIL(M2D_ERR_RANGE, "m2d_conv1d_fwd_k4: tensor exceeds 2^31 elements");
  const int ng = k4_groups(ks, pad), P = pad + k4_shift(pad);
  M2dGemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = Cout;
  p.N = B * Lout;
  p.K = Cin * ks;   // real taps (reported work); the kernel multiplies Cin * ng * 4
  p.nhi = Cin;
  p.kdiv = 4 * ng;
  p.k4_ng = ng;
  p.k4_pair = k4_paired(Cin, ks, pad);
  p.k4_shift = k4_shift(pad);
  p.k4_nlast = k4_nlast(ks, pad);
  p.phases = 1;
  p.A.base = w_k4;
  p.A.nbytes = m2d_extent_bytes((long long)m2d_conv1d_k4_packed_elems(Cout, Cin, ks, pad));
  p.A.nrows = Cout;
  p.A.r_lo_stride = Cout * 4;   // floats per tap group of the packed image
  M2dOperand& b = p.B;
  b.base = x;
  b.nbytes = m2d_extent_bytes((long long)B * Cin * L);
  b.nrows = p.N;
  b.rdiv = Lout;
  b.rdiv_inv = 1.f / (float)Lout;
  b.r_hi_stride = Cin * L;
  b.r_lo_stride = stride;
  b.r_off = -P;
  b.r_pos_mul = stride;
  b.r_pos_off = -P;
  b.k_hi_stride = L;
  b.lim = L;
  m2d_outmap_plain(p.O, y, Lout, 1);
  p.O.cdiv = Lout;
  p.O.cdiv_inv = 1.f / (float)Lout;
  p.O.c_hi_stride = Cout * Lout;
  p.O.c_lo_stride = 1;
  p.O.bias = bias;
  p.O.bias_mode = bias ? 1 : 0;
  p.O.act = act;
  p.O.slope = slope;
  p.O.residual = residual;
  p.O.sum_out = sum_out;
  p.O.mask = out_mask;
  p.O.mask_slope = out_mask_slope;
  if (stats) {   // the statistics partials first, split-K slabs (one-launch form only) behind them
    const size_t st = m2d_rowstats_bytes(p.M, p.N);
    if (!ws || ws_bytes < st) M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_conv1d_fwd_k4: no room for the statistics partials");
    p.O.row_part = (float*)ws;
    p.O.row_sums = stats;
    ws = (char*)ws + st;
    ws_bytes -= st;
  }
  return m2d_conv_k4_launch(p, /*allow_split=*/true, ws, ws_bytes, (hipStream_t)stream, "m2d_conv1d_fwd");
}

}  // extern "C"

extern "C" {

// The first conv of an audio encoder applied to the windows of a padded track WITHOUT writing the
// windows: replaces utils.slice_audio_batch (utils.py:329-353) + nn.Conv1d(1, Cout, ...) on the
// (B*T, 1, window) slices (phase3/archis/default.py:27-28,64,90,117). track: (B, S) floats, window t of
// track b = track[b, t*hop : t*hop + window]; y: (B*T, Cout, Lout).
int m2d_conv1d_fwd_windows(const float* track, int B, int S, int T, int hop, int window, const float* w,
                           const float* bias, float* y, int Cout, int ks, int stride, int pad, int act, float slope,
                           double* stats, void* ws, size_t ws_bytes, void* stream) {
  if (B <= 0 || T <= 0 || hop <= 0 || window <= 0 || (long long)(T - 1) * hop + window > S)
    M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_fwd_windows: windows do not fit the track (S=%d T=%d hop=%d window=%d)", S, T,
             hop, window);
  if (!fits_i32((long long)B * S) || !fits_i32((long long)B * T * window))
    M2D_FAIL(M2D_ERR_RANGE, "m2d_conv1d_fwd_windows: tensor exceeds 2^31 elements");
  const M2dWinView wv = {T, S, hop};
  return conv1d_fwd_impl(track, w, nullptr, bias, y, B * T, 1, window, Cout, ks, stride, pad, act, slope, nullptr,
                         nullptr, 0.f, ws, ws_bytes, stream, &wv, stats);
}

// Replaces the input-gradient half of convolution_backward (autograd of nn.Conv1d), the
// op the gradient penalty differentiates a second time (losses.py:40-44).
// `w_packed` (optional): the backward image (Cout, ks, Cin) of w from m2d_conv1d_pack_weights.
// `dy_mask` (optional, shape of dy): dy is read as dy * (mask>0 ? 1 : dy_mask_slope).
static int conv1d_bwd_data_impl(const float* dy, const float* w, const float* w_packed, float* dx, int B, int Cin, int L,
                                int Cout, int ks, int stride, int pad, const float* dy_mask, float dy_mask_slope,
                                const float* out_mask, float out_mask_slope, void* ws, size_t ws_bytes, void* stream,
                                const float* residual, int mask_batch = 0) {
  const int Lout = conv_out_len(L, ks, stride, pad);
  if (B <= 0 || Cin <= 0 || Cout <= 0 || Lout <= 0)
    M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_bwd_data: bad shape");
  // shared mask: out_mask holds mask_batch samples, sample n of dx reads mask sample n - mask_batch once n >= mask_batch
  if (mask_batch < 0 || (mask_batch > 0 && (!out_mask || 2 * mask_batch < B)))
    M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_bwd_data: a shared mask needs an out_mask of >= B / 2 samples (mask_batch=%d, B=%d)", mask_batch, B);
  const unsigned mask_wrap = (mask_batch > 0 && mask_batch < B) ? (unsigned)((long long)mask_batch * Cin * L) : 0u;
  if (!fits_i32((long long)B * Cin * L) || !fits_i32((long long)B * Cout * Lout) ||
      !fits_i32((long long)Cout * Cin * ks))
    M2D_FAIL(M2D_ERR_RANGE, "m2d_conv1d_bwd_data: tensor exceeds 2^31 elements");
  if (m2d_thin_applicable(Cin, Cout, ks, stride) && !out_mask && !residual)
    return m2d_thin_bwd_data(dy, w, dx, B, L, Cout, ks, stride, pad, Lout, dy_mask, dy_mask_slope,
                             (hipStream_t)stream);
  M2dGemmParams p;
  memset(&p, 0, sizeof(p));
  if (Lout == 1 && pad == 0 && L == ks) {
    // full-length kernel (fconv / l6 / last encoder conv): dx[n,(ci,kk)] = sum_co dy[n,co] W[co,(ci,kk)]
    p.M = B;
    p.N = Cin * ks;
    p.K = Cout;
    p.nhi = 1;
    p.kdiv = Cout;
    p.phases = 1;
    m2d_operand_plain(p.A, dy, B, Cout, 1, (long long)B * Cout);
    p.A.mask = dy_mask;
    p.A.mask_slope = dy_mask_slope;
    m2d_operand_plain(p.B, w, p.N, 1, Cin * ks, (long long)Cout * Cin * ks);
    m2d_outmap_plain(p.O, dx, Cin * ks, 1);
    p.O.mask = out_mask;
    p.O.mask_wrap = mask_wrap;
    p.O.mask_slope = out_mask_slope;
    p.O.residual = residual;
    p.O.mask_last = 1;
    return m2d_gemm_launch(p, true, false, true, ws, ws_bytes, (hipStream_t)stream, "m2d_conv1d_bwd_data");
  }
  if (bwd_subpixel(Cin, Cout, ks, stride, dy_mask != nullptr)) {
    // dx[n, ci, s q + r - pad] = sum_{t, co} w[co, ci, r + s t] dy[n, co, q - t]: rows (ci, r), K = (t, co), columns (n, q)
    const int s = stride, nt = (ks + s - 1) / s;
    const bool tall = subpixel_tall(Cin, Cout, ks, s) && !dy_mask;  // (a masked dy runs on the register-staging kernel)
    const size_t pb = subpixel_bytes(Cout, Cin, ks, s);
    if (!ws || ws_bytes < pb) M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_conv1d_bwd_data: workspace too small for the sub-pixel weights");
    {
      const size_t total = (size_t)nt * Cout * Cin * s;
      unsigned blocks = (unsigned)((total + 255) / 256);
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(m2d_pack_weights_subpixel_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (float*)ws,
                         Cout, Cin, ks, s, nt, tall ? 1 : 0);
    }
    const float* wsp = (const float*)ws;
    ws = (char*)ws + pb;
    ws_bytes -= pb;
    // q range over all phases: s q + r - pad in [0, L) for some r
    const int qmin = pad / s;                              // r = s - 1 reaches furthest left: ceil((pad - s + 1) / s)
    const int qlo = (pad - (s - 1) + s - 1) / s > 0 ? (pad - (s - 1) + s - 1) / s : 0;
    const int qhi = (L - 1 + pad) / s;                     // r = 0 reaches furthest right
    (void)qmin;
    const int nq = qhi - qlo + 1;
    p.M = Cin * s;
    p.N = B * nq;
    p.K = Cout * nt;
    p.nhi = nt;
    p.kdiv = Cout;
    p.lo_outer = 1;
    p.phases = 1;
    m2d_operand_plain(p.A, wsp, p.M, 1, p.M, (long long)nt * Cout * p.M);   // A(m, k = (t, co)) = wsp[(t * Cout + co) * M + m]
    p.A.k_hi_stride = Cout * p.M;
    M2dOperand& b = p.B;
    b.base = dy;
    b.nbytes = m2d_extent_bytes((long long)B * Cout * Lout);
    b.mask = dy_mask;
    b.mask_slope = dy_mask_slope;
    b.nrows = p.N;
    b.rdiv = nq;
    b.rdiv_inv = 1.f / (float)nq;
    b.r_hi_stride = Cout * Lout;
    b.r_lo_stride = 1;
    b.r_off = qlo;
    b.r_pos_mul = 1;
    b.r_pos_off = qlo;
    b.k_hi_stride = -1;
    b.k_lo_stride = Lout;
    b.k_pos_hi = -1;
    b.k_pos_lo = 0;
    b.k_safe_lo = 0;
    b.k_safe_hi = 0x7fffffff;
    b.lim = Lout;
    // out(m = (ci, r), col = (n, q)) = dx[n * Cin * L + ci * L + s * q + r - pad], written iff s q + r - pad in [0, L)
    m2d_outmap_plain(p.O, dx, L, 1);
    p.O.m_div = s;
    p.O.m_lo_stride = 1;
    p.O.m_pos_mul = 1;
    p.O.cdiv = nq;
    p.O.cdiv_inv = b.rdiv_inv;
    p.O.c_hi_stride = Cin * L;
    p.O.c_lo_stride = s;
    p.O.c_off = s * qlo - pad;
    p.O.c_pos_mul = s;
    p.O.c_pos_off = s * qlo - pad;
    p.O.c_lim = L;
    p.O.mask = out_mask;
    p.O.mask_wrap = mask_wrap;
    p.O.mask_slope = out_mask_slope;
    p.O.residual = residual;
    p.O.mask_last = 1;
    p.O.quad = s == 4 ? 1 : 0;  // M = 4 Cin
    p.plan_kind = M2D_PLAN_BWD_DATA;
    p.tall_last_rb = tall ? ks - s * (nt - 1) : 0;
    // reported work: the real taps (as the polyphase form counts them), not the padded K
    for (int r = 0; r < s; ++r) {
      const int taps = r < ks ? (ks - r + s - 1) / s : 0;
      const int q0 = r >= pad ? 0 : (pad - r + s - 1) / s;
      const int top = L - 1 + pad - r;
      const int nqr = top >= 0 ? (top / s - q0 + 1) : 0;
      if (nqr > 0) p.work_flops += 2.0 * Cin * (double)B * nqr * Cout * taps;
    }
    return m2d_gemm_launch(p, /*a_kfast=*/false, /*b_kfast=*/false, /*allow_split=*/false, ws, ws_bytes,
                           (hipStream_t)stream, "m2d_conv1d_bwd_data");
  }
  if (!w_packed) {
    const size_t pb = pack_bytes(Cout, Cin, ks);
    if (!ws || ws_bytes < pb)
      M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_conv1d_bwd_data: workspace too small to pack the weights");
    const int rc = pack_weights(w, (float*)ws, nullptr, Cout, Cin, ks, (hipStream_t)stream);
    if (rc) return rc;
    w_packed = (const float*)ws;
    ws = (char*)ws + pb;
    ws_bytes -= pb;
  }
  if (stride == 1 && Lout == L && (mask_wrap % 4u) == 0) {
    // dx[n, ci, j] = sum_{t', co} wb[co, ks - 1 - t', ci] dy[n, co, j - pad + t']: the "same" convolution of dy with the
    // flipped taps of the (Cout, ks, Cin) image - the TemporalBlock kernel with the roles of the channels swapped
    const int nt = m2d_tcn_conv_tile(B, Cout, L, Cin, ks, stride, pad);
    if (nt > 0) {
      M2dTcnConv t;
      memset(&t, 0, sizeof(t));
      t.x = dy;
      t.x_mask = dy_mask;
      t.x_mask_slope = dy_mask_slope;
      t.wimg = w_packed;
      t.out = dx;
      t.out_mask = out_mask;
      t.out_mask_slope = out_mask_slope;
      t.out_mask_wrap = mask_wrap;
      t.residual = residual;
      t.mask_last = 1;
      t.tap_rev = 1;
      t.B = B;
      t.Cin = Cout;
      t.L = L;
      return m2d_tcn_conv_launch(t, ks, nt, (hipStream_t)stream, "m2d_conv1d_bwd_data");
    }
  }
  // dx[n, ci, s*q + r - pad] = sum_{t, co} wb[ci, r + s*t, co] * dy[n, co, q - t] per output phase r:
  // K = (t, co), hi = t (taps(r) of them, resolved on the device), lo = co.
  p.bwd_data = 1;
  p.plan_kind = stride > 1 ? M2D_PLAN_BWD_DATA : M2D_PLAN_GENERAL;
  p.phases = stride;
  p.ph_ks = ks;
  p.ph_cout = Cout;
  p.ph_pad = pad;
  p.ph_L = L;
  p.ph_batch = B;
  p.M = Cin;
  p.kdiv = Cout;
  p.lo_outer = 1;
  p.nhi = (ks + stride - 1) / stride;
  // widest phase: q in [qmin, (L-1+pad-r)/s]
  int nq_max = 0;
  for (int r = 0; r < stride; ++r) {
    const int qmin = r >= pad ? 0 : (pad - r + stride - 1) / stride;
    const int top = L - 1 + pad - r;
    const int nq = top >= 0 ? (top / stride - qmin + 1) : 0;
    if (nq > nq_max) nq_max = nq;
  }
  p.N = B * nq_max;
  p.K = Cout * p.nhi;
  // K-major weight image (Cout, ks, Cin): A(m = ci, k = (t,co)) = wk[co*ks*Cin + (r + s*t)*Cin + ci], rows contiguous
  // (r*Cin added per phase on the device)
  m2d_operand_plain(p.A, w_packed, Cin, 1, ks * Cin, (long long)Cout * Cin * ks);
  p.A.k_hi_stride = stride * Cin;
  p.ph_a_step = Cin;
  // B(k = (t,co), col = (n,q)) = dy[n*Cout*Lout + co*Lout + q - t], valid iff 0 <= q - t < Lout
  M2dOperand& b = p.B;
  b.base = dy;
  b.nbytes = m2d_extent_bytes((long long)B * Cout * Lout);
  b.mask = dy_mask;
  b.mask_slope = dy_mask_slope;
  b.nrows = p.N;
  b.rdiv = nq_max;
  b.rdiv_inv = 1.f / (float)nq_max;
  b.r_hi_stride = Cout * Lout;
  b.r_lo_stride = 1;
  b.r_pos_mul = 1;
  b.k_hi_stride = -1;
  b.k_lo_stride = Lout;
  b.k_pos_hi = -1;
  b.k_pos_lo = 0;
  b.k_safe_lo = 0;
  b.k_safe_hi = 0x7fffffff;
  b.lim = Lout;
  // out(m = ci, col = (n,q)) = dx[n*Cin*L + ci*L + s*q + r - pad]
  m2d_outmap_plain(p.O, dx, L, 1);
  p.O.c_hi_stride = Cin * L;
  p.O.c_lo_stride = stride;
  p.O.c_pos_mul = stride;
  p.O.c_lim = L;
  p.O.mask = out_mask;
  p.O.mask_wrap = mask_wrap;
  p.O.mask_slope = out_mask_slope;
  p.O.residual = residual;
  p.O.mask_last = 1;
  if (stride == 1) {
    // single phase: resolve the phase parameters here (r = 0, taps = ks, q in [pad, L-1+pad]) so the
    // launch is an ordinary GEMM and may be split along K (the TCN critic's small grids need it)
    p.bwd_data = 0;
    p.phases = 1;
    p.N = B * L;
    p.nhi = ks;
    p.K = Cout * ks;
    b.nrows = p.N;
    b.rdiv = L;
    b.rdiv_inv = 1.f / (float)L;
    b.r_off = pad;
    b.r_pos_off = pad;
    p.O.cdiv = L;
    p.O.cdiv_inv = b.rdiv_inv;
    p.O.c_off = 0;
    p.O.c_pos_off = 0;
    p.O.c_lim = 0;  // every column (n, j), j < L, is an output position: no window test (16-byte epilogue rows)
    return m2d_gemm_launch(p, /*a_kfast=*/false, /*b_kfast=*/false, /*allow_split=*/true, ws, ws_bytes,
                           (hipStream_t)stream, "m2d_conv1d_bwd_data");
  }
  return m2d_gemm_launch(p, /*a_kfast=*/false, /*b_kfast=*/false, /*allow_split=*/false, ws, ws_bytes,
                         (hipStream_t)stream, "m2d_conv1d_bwd_data");
}

int m2d_conv1d_bwd_data(const float* dy, const float* w, const float* w_packed, float* dx, int B, int Cin, int L,
                        int Cout, int ks, int stride, int pad, const float* dy_mask, float dy_mask_slope,
                        const float* out_mask, float out_mask_slope, void* ws, size_t ws_bytes, void* stream) {
  return conv1d_bwd_data_impl(dy, w, w_packed, dx, B, Cin, L, Cout, ks, stride, pad, dy_mask, dy_mask_slope, out_mask,
                              out_mask_slope, ws, ws_bytes, stream, nullptr);
}

// dx = out_mask * (conv^T(dy * dy_mask, W) + residual): the input gradient of a layer whose input also feeds a
// skip connection (TemporalBlock: d x = d out + conv1^T(...), phase3/archis/default.py:207-210) - the skip
// gradient is added in the epilogue instead of by a separate accumulation pass. residual: shape of dx.
int m2d_conv1d_bwd_data_res(const float* dy, const float* w, const float* w_packed, float* dx, int B, int Cin, int L,
                            int Cout, int ks, int stride, int pad, const float* dy_mask, float dy_mask_slope,
                            const float* residual, const float* out_mask, float out_mask_slope, void* ws,
                            size_t ws_bytes, void* stream) {
  return conv1d_bwd_data_impl(dy, w, w_packed, dx, B, Cin, L, Cout, ks, stride, pad, dy_mask, dy_mask_slope, out_mask,
                              out_mask_slope, ws, ws_bytes, stream, residual);
}

// m2d_conv1d_bwd_data over a batch whose two halves share ONE set of activation masks: out_mask holds `mask_batch`
// samples (B / 2 <= mask_batch <= B) and sample n >= mask_batch of dx reads mask sample n - mask_batch. The audio branch
// of the phase-3 critic (phase3/archis/default.py:312-319) sees the same audio in the penalty pass and in the real /
// fake passes, so the penalty's first backward (losses.py:40-44) and the loss backward (phase3/train.py:215) travel
// through identical ReLU masks: one launch over 2B gradient rows per layer instead of two over B.
int m2d_conv1d_bwd_data_shared_mask(const float* dy, const float* w, const float* w_packed, float* dx, int B, int Cin,
                                    int L, int Cout, int ks, int stride, int pad, const float* out_mask,
                                    float out_mask_slope, int mask_batch, void* ws, size_t ws_bytes, void* stream) {
  if (m2d_thin_applicable(Cin, Cout, ks, stride))
    M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_bwd_data_shared_mask: not for the thin (Cin = 1) layer");
  return conv1d_bwd_data_impl(dy, w, w_packed, dx, B, Cin, L, Cout, ks, stride, pad, nullptr, 0.f, out_mask,
                              out_mask_slope, ws, ws_bytes, stream, nullptr, mask_batch);
}

// Replaces the weight-gradient half of convolution_backward. K = (sample, position) is the long
// dimension (hi = n, lo = l), so the launch is split-K with a deterministic slab reduction.
// `dbias` (optional, Cout floats): the bias gradient sum_{n,l} dy[n,co,l] (masked like dy) from the
// same launch - one all-ones column appended to the x operand - instead of a second pass over dy.
static int conv1d_bwd_weight_impl(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int L,
                                  int Cout, int ks, int stride, int pad, const float* dy_mask, float dy_mask_slope,
                                  void* ws, size_t ws_bytes, void* stream, const M2dWinView* wv,
                                  int bias_from_sample = 0) {
  const int Lout = conv_out_len(L, ks, stride, pad);
  if (bias_from_sample < 0 || bias_from_sample > B) M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_bwd_weight: bad bias_from_sample");
  if (bias_from_sample > 0 && dbias && (Lout < M2D_BK || m2d_thin_applicable(Cin, Cout, ks, stride)))
    M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_bwd_weight: bias_from_sample needs >= %d output positions and Cin > 1", M2D_BK);
  if (B <= 0 || Cin <= 0 || Cout <= 0 || Lout <= 0)
    M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_bwd_weight: bad shape");
  if (!fits_i32((long long)B * Cin * L) || !fits_i32((long long)B * Cout * Lout))
    M2D_FAIL(M2D_ERR_RANGE, "m2d_conv1d_bwd_weight: tensor exceeds 2^31 elements");
  if (m2d_thin_applicable(Cin, Cout, ks, stride))
    return m2d_thin_bwd_weight(x, dy, dw, dbias, B, L, Cout, ks, stride, pad, Lout, dy_mask, dy_mask_slope, ws,
                               ws_bytes, wv, (hipStream_t)stream);
  if (wv && (Cin != 1 || Lout < M2D_BK))
    M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_bwd_weight_windows: single-channel convs with >= %d output positions only", M2D_BK);
  if (!wv && Lout == L && m2d_tcn_wgrad_applicable(B, Cin, L, Cout, ks, stride, pad)) {
    // the pose critic's TemporalBlock convolutions: csrc/tcn.hip
    M2dTcnWgrad t;
    memset(&t, 0, sizeof(t));
    t.x = x;
    t.dy = dy;
    t.dy_mask = dy_mask;
    t.dy_mask_slope = dy_mask_slope;
    t.dw = dw;
    t.dbias = dbias;
    t.bias_from = bias_from_sample;
    t.B = B;
    t.Cin = Cin;
    t.L = L;
    return m2d_tcn_wgrad_launch(t, ks, ws, ws_bytes, (hipStream_t)stream, "m2d_conv1d_bwd_weight");
  }
  M2dGemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = Cout;
  p.N = Cin * ks;
  p.K = B * Lout;
  p.phases = 1;
  M2dOperand& b = p.B;
  b.base = x;
  b.nbytes = m2d_extent_bytes((long long)B * Cin * L);
  b.nrows = p.N;
  b.rdiv = ks;
  b.rdiv_inv = 1.f / (float)ks;
  b.r_hi_stride = L;
  b.r_lo_stride = 1;
  b.r_off = -pad;
  b.r_pos_mul = 1;
  b.r_pos_off = -pad;
  b.lim = L;
  bool a_kfast = true;
  if (Lout >= M2D_BK) {
    // K = (n, l): hi = sample, lo = position (contiguous in dy)
    p.nhi = B;
    p.kdiv = Lout;
    // A(m = co, k = (n,l)) = dy[n*Cout*Lout + co*Lout + l]
    m2d_operand_plain(p.A, dy, Cout, Lout, 1, (long long)B * Cout * Lout);
    p.A.k_hi_stride = Cout * Lout;
    // B(k = (n,l), col = (ci,kk)) = x[n*Cin*L + ci*L + l*s - pad + kk]
    b.k_hi_stride = Cin * L;
    b.k_lo_stride = stride;
    b.k_pos_hi = 0;
    b.k_pos_lo = stride;
    if (wv && wv->T > 0) {  // sample n = (track, window t) at track * S + t * hop
      b.nbytes = m2d_extent_bytes((long long)(B / wv->T) * wv->S);
      b.k_hi_stride = wv->hop;
      b.kdiv2 = wv->T;
      b.k_hi2_stride = wv->S;
    }
    // positions l whose every tap l*s - pad + kk, kk in [0, ks), lies in [0, L)
    b.k_safe_lo = (pad + stride - 1) / stride;
    b.k_safe_hi = (L - ks + pad) >= 0 ? (L - ks + pad) / stride + 1 : 0;
    p.plan_kind = M2D_PLAN_BWD_WEIGHT;
  } else {
    // short outputs (the encoders' deep layers, full-length kernels): K = (l, n), hi = position,
    // lo = sample, so no chunk is mostly padding; dy is read row-fast (lanes along co, Lout apart)
    // and the window depends on the chunk only
    p.nhi = Lout;
    p.kdiv = B;
    a_kfast = false;
    m2d_operand_plain(p.A, dy, Cout, Lout, Cout * Lout, (long long)B * Cout * Lout);
    p.A.k_hi_stride = 1;
    b.k_hi_stride = stride;
    b.k_lo_stride = Cin * L;
    b.k_pos_hi = stride;
    b.k_pos_lo = 0;
    b.k_safe_lo = 0;
    b.k_safe_hi = 0x7fffffff;
  }
  p.A.mask = dy_mask;
  p.A.mask_slope = dy_mask_slope;
  m2d_outmap_plain(p.O, dw, Cin * ks, 1);
  if (dbias) {
    // column Cin*ks of the x operand reads as ones: dbias[co] = sum_k dy[co, k]
    p.N = Cin * ks + 1;
    b.ones_row_p1 = Cin * ks + 1;
    b.ones_from_hi = bias_from_sample;
    p.O.col_out = dbias;
    p.O.redirect_col_p1 = Cin * ks + 1;
  }
  return m2d_gemm_launch(p, a_kfast, /*b_kfast=*/false, /*allow_split=*/true, ws, ws_bytes, (hipStream_t)stream,
                         "m2d_conv1d_bwd_weight");
}

int m2d_conv1d_bwd_weight(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int L,
                          int Cout, int ks, int stride, int pad, const float* dy_mask, float dy_mask_slope,
                          void* ws, size_t ws_bytes, void* stream) {
  return conv1d_bwd_weight_impl(x, dy, dw, dbias, B, Cin, L, Cout, ks, stride, pad, dy_mask, dy_mask_slope, ws,
                                ws_bytes, stream, nullptr);
}

// The same with the bias gradient summed over the samples [bias_from_sample, B) only. One launch then serves a
// batch whose first rows pair SECOND-order operands (x := the penalty's forward-mode tangent, dy := the first
// backward's input gradient; they feed dW but no bias) with ordinary (x, dy) rows behind them - the critic
// iteration's two weight-gradient launches per layer (losses.py:40-44 double backward + loss backward) as one.
int m2d_conv1d_bwd_weight_from(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int L,
                               int Cout, int ks, int stride, int pad, const float* dy_mask, float dy_mask_slope,
                               int bias_from_sample, void* ws, size_t ws_bytes, void* stream) {
  return conv1d_bwd_weight_impl(x, dy, dw, dbias, B, Cin, L, Cout, ks, stride, pad, dy_mask, dy_mask_slope, ws,
                                ws_bytes, stream, nullptr, bias_from_sample);
}

// Weight (and bias) gradient of m2d_conv1d_fwd_windows; dy: (B*T, Cout, Lout). Workspace as for the
// dense call on (B*T, 1, window).
int m2d_conv1d_bwd_weight_windows(const float* track, int B, int S, int T, int hop, int window, const float* dy,
                                  float* dw, float* dbias, int Cout, int ks, int stride, int pad,
                                  const float* dy_mask, float dy_mask_slope, void* ws, size_t ws_bytes, void* stream) {
  if (B <= 0 || T <= 0 || hop <= 0 || window <= 0 || (long long)(T - 1) * hop + window > S)
    M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_bwd_weight_windows: windows do not fit the track");
  if (!fits_i32((long long)B * S) || !fits_i32((long long)B * T * window))
    M2D_FAIL(M2D_ERR_RANGE, "m2d_conv1d_bwd_weight_windows: tensor exceeds 2^31 elements");
  const M2dWinView wv = {T, S, hop};
  return conv1d_bwd_weight_impl(track, dy, dw, dbias, B * T, 1, window, Cout, ks, stride, pad, dy_mask, dy_mask_slope,
                                ws, ws_bytes, stream, &wv);
}

// which: 0 forward, 1 backward-data, 2 backward-weight. Includes the room forward / backward-data
// need to pack the weights when the caller passes no `w_packed`.
size_t m2d_conv1d_workspace_bytes(int which, int B, int Cin, int L, int Cout, int ks, int stride, int pad) {
  const int Lout = conv_out_len(L, ks, stride, pad);
  if (Lout <= 0) return 0;
  if (m2d_thin_applicable(Cin, Cout, ks, stride))
    return which == 2 ? m2d_thin_bwd_weight_ws(B, Cout, ks, Lout) : (which == 0 ? m2d_thin_fwd_stats_ws(B, Lout) : 0);
  if (which == 0 && m2d_thin_long_applicable(Cin, Cout, ks, stride, pad, L)) {
    // (a masked launch of this layer goes through the engine: room for both)
    const size_t a = m2d_gemm_plan(Cout, B * Lout, m2d_chunks(Cin, ks), 1, true).ws_bytes + m2d_rowstats_bytes(Cout, B * Lout);
    const size_t t = m2d_thin_long_stats_ws(B, Lout);
    return a > t ? a : t;
  }
  if (which == 0) {
    // split-K slabs behind the per-tile partials of the epilogue statistics
    if (Lout == 1 && pad == 0 && L == ks) {
      const size_t a = m2d_gemm_plan(Cout, B, m2d_chunks(1, Cin * ks), 1, true).ws_bytes, st = m2d_rowstats_bytes(Cout, B);
      return a + st;
    }
    const bool packed = conv_uses_packed(Cin);
    const int nch = packed ? m2d_chunks(ks, Cin) : m2d_chunks(Cin, ks);
    const size_t a = m2d_gemm_plan(Cout, B * Lout, nch, 1, true, packed ? fwd_tile_penalty(stride) : 1.f).ws_bytes;
    const size_t st = m2d_rowstats_bytes(Cout, B * Lout);
    return a + st + (packed ? pack_bytes(Cout, Cin, ks) : 0);
  }
  if (which == 1) {
    if (Lout == 1 && pad == 0 && L == ks) return m2d_gemm_plan(B, Cin * ks, m2d_chunks(1, Cout), 1, true).ws_bytes;
    if (stride == 1) return m2d_gemm_plan(Cin, B * L, m2d_chunks(ks, Cout), 1, true).ws_bytes + pack_bytes(Cout, Cin, ks);
    if (bwd_subpixel(Cin, Cout, ks, stride)) return subpixel_bytes(Cout, Cin, ks, stride) + pack_bytes(Cout, Cin, ks);
    return pack_bytes(Cout, Cin, ks);
  }
  // sized for the launch with the bias column (one more column), which is never smaller
  const int nch = Lout >= M2D_BK ? m2d_chunks(B, Lout) : m2d_chunks(Lout, B);
  const int kind = Lout >= M2D_BK ? M2D_PLAN_BWD_WEIGHT : M2D_PLAN_GENERAL;
  const size_t a = m2d_gemm_plan(Cout, Cin * ks, nch, 1, true, 1.0, kind).ws_bytes;
  const size_t c = m2d_gemm_plan(Cout, Cin * ks + 1, nch, 1, true, 1.0, kind).ws_bytes;
  const size_t t = (Lout == L && m2d_tcn_wgrad_applicable(B, Cin, L, Cout, ks, stride, pad)) ? m2d_tcn_wgrad_ws_bytes(B, Cin, L, ks) : 0;
  return (a > c ? a : c) > t ? (a > c ? a : c) : t;
}

// Dense row-major GEMMs behind nn.Linear (phase3/archis/default.py:153,161,176-177,256-257;
// phase1/archis/residual.py:11,19,35,41) and the GRU input projections:
//   mode 0 (NT): C[M,N] = A[M,K] * B[N,K]^T (+ bias[N], act)         y  = x W^T + b
//   mode 1 (NN): C[M,N] = A[M,K] * B[K,N]                              dx = dy W
//   mode 2 (TN): C[M,N] = A[K,M]^T * B[K,N]                            dW = dy^T x
// a_mask (shape of A) / out_mask (shape of C) fuse the activation derivative as in conv1d.
static int gemm_impl(int mode, const float* a, int lda, const float* b, int ldb, const float* bias, float* c, int ldc,
                     int M, int N, int K, int act, float slope, const float* a_mask, float a_mask_slope,
                     const float* out_mask, float out_mask_slope, void* ws, size_t ws_bytes, void* stream) {
  if (M < 0 || N < 0 || K < 0 || mode < 0 || mode > 2) M2D_FAIL(M2D_ERR_ARG, "m2d_gemm: bad arguments");
  // leading dimensions (row pitch in elements) of the stored matrices; 0 = dense
  const int a_cols = mode == 2 ? M : K, b_cols = mode == 0 ? K : N;
  if (lda == 0) lda = a_cols;
  if (ldb == 0) ldb = b_cols;
  if (ldc == 0) ldc = N;
  if (lda < a_cols || ldb < b_cols || ldc < N) M2D_FAIL(M2D_ERR_ARG, "m2d_gemm: leading dimension smaller than the row");
  if (!fits_i32((long long)M * K) || !fits_i32((long long)N * K) || !fits_i32((long long)M * N))
    M2D_FAIL(M2D_ERR_RANGE, "m2d_gemm: matrix exceeds 2^31 elements");
  M2dGemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = M;
  p.N = N;
  p.K = K;
  p.nhi = K > 0 ? 1 : 0;
  p.kdiv = K > 0 ? K : 1;
  p.phases = 1;
  bool akf = true, bkf = true;
  // an empty contraction still runs the epilogue (bias / activation of zero): give the
  // descriptors a non-empty extent, nothing is read through them
  const long long a_rows = mode == 2 ? K : M, b_rows = mode == 0 ? N : K;
  const long long ea = K > 0 ? (a_rows - 1) * lda + a_cols : 1, eb = K > 0 ? (b_rows - 1) * ldb + b_cols : 1;
  if (!fits_i32((long long)(M - 1) * ldc + N)) M2D_FAIL(M2D_ERR_RANGE, "m2d_gemm: output exceeds 2^31 elements");
  if (mode == 0) {
    m2d_operand_plain(p.A, a, M, lda, 1, ea);
    m2d_operand_plain(p.B, b, N, ldb, 1, eb);
    akf = true;
    bkf = true;
  } else if (mode == 1) {
    m2d_operand_plain(p.A, a, M, lda, 1, ea);
    m2d_operand_plain(p.B, b, N, 1, ldb, eb);
    akf = true;
    bkf = false;
  } else {
    m2d_operand_plain(p.A, a, M, 1, lda, ea);
    m2d_operand_plain(p.B, b, N, 1, ldb, eb);
    akf = false;
    bkf = false;
  }
  p.A.mask = a_mask;
  p.A.mask_slope = a_mask_slope;
  m2d_outmap_plain(p.O, c, ldc, 1);
  p.O.bias = bias;
  p.O.bias_mode = bias ? 2 : 0;
  p.O.act = act;
  p.O.slope = slope;
  p.O.mask = out_mask;
  p.O.mask_slope = out_mask_slope;
  return m2d_gemm_launch(p, akf, bkf, true, ws, ws_bytes, (hipStream_t)stream, "m2d_gemm");
}

int m2d_gemm(int mode, const float* a, const float* b, const float* bias, float* c, int M, int N, int K,
             int act, float slope, const float* a_mask, float a_mask_slope, const float* out_mask,
             float out_mask_slope, void* ws, size_t ws_bytes, void* stream) {
  return gemm_impl(mode, a, 0, b, 0, bias, c, 0, M, N, K, act, slope, a_mask, a_mask_slope, out_mask, out_mask_slope,
                   ws, ws_bytes, stream);
}

// m2d_gemm on sub-matrices: lda / ldb / ldc are the row pitches (elements) of the STORED a / b / c (0 = dense);
// a_mask shares lda, out_mask shares ldc. Lets the critic head read / write column blocks of wider buffers in
// place (the concatenated (B, 200) code of phase3/archis/default.py:266-269 is never assembled by a copy).
int m2d_gemm_ld(int mode, const float* a, int lda, const float* b, int ldb, const float* bias, float* c, int ldc,
                int M, int N, int K, int act, float slope, const float* a_mask, float a_mask_slope,
                const float* out_mask, float out_mask_slope, void* ws, size_t ws_bytes, void* stream) {
  return gemm_impl(mode, a, lda, b, ldb, bias, c, ldc, M, N, K, act, slope, a_mask, a_mask_slope, out_mask,
                   out_mask_slope, ws, ws_bytes, stream);
}

size_t m2d_gemm_workspace_bytes(int mode, int M, int N, int K) {
  (void)mode;
  return m2d_gemm_plan(M, N, m2d_chunks(K > 0 ? 1 : 0, K > 0 ? K : 1), 1, true).ws_bytes;
}

}  // extern "C"
