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

static inline int conv_out_len(int L, int ks, int stride, int pad) {
  const int span = L + 2 * pad - ks;
  return span < 0 ? 0 : span / stride + 1;
}

static inline bool fits_i32(long long v) { return v >= 0 && v < 2147483647LL; }

// conv1d_thin.hip: vector-ALU kernels for Cin = 1, k = 25, stride 4 (HBM-bound layers)
bool m2d_thin_applicable(int Cin, int Cout, int ks, int stride);
size_t m2d_thin_bwd_weight_ws(int B, int Cout, int ks, int Lout);
int m2d_thin_fwd(const float* x, const float* w, const float* bias, float* y, int B, int L, int Cout, int ks,
                 int stride, int pad, int Lout, int act, float slope, const float* out_mask, float out_mask_slope,
                 hipStream_t stream);
int m2d_thin_bwd_data(const float* dy, const float* w, float* dx, int B, int L, int Cout, int ks, int stride,
                      int pad, int Lout, const float* dy_mask, float dy_mask_slope, hipStream_t stream);
int m2d_thin_bwd_weight(const float* x, const float* dy, float* dw, int B, int L, int Cout, int ks, int stride,
                        int pad, int Lout, const float* dy_mask, float dy_mask_slope, void* ws, size_t ws_bytes,
                        hipStream_t stream);

static void fill_fwd(M2dGemmParams& p, const float* x, const float* w, float* y, int B, int Cin,
                     int L, int Cout, int ks, int stride, int pad, int Lout) {
  memset(&p, 0, sizeof(p));
  p.M = Cout;
  p.N = B * Lout;
  p.K = Cin * ks;
  p.phases = 1;
  // A(m = co, k = (ci,kk)) = W[co*Cin*ks + k]
  m2d_operand_plain(p.A, w, Cout, Cin * ks, 1, (long long)Cout * Cin * ks);
  // B(k = (ci,kk), col = (n,l)) = x[n*Cin*L + ci*L + l*s - p + kk]
  M2dOperand& b = p.B;
  memset(&b, 0, sizeof(b));
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
  b.kdiv = ks;
  b.kdiv_inv = 1.f / (float)ks;
  b.k_hi_stride = L;
  b.k_lo_stride = 1;
  b.k_pos_mul = 1;
  b.lim = L;
  // out(m = co, col = (n,l)) = y[n*Cout*Lout + co*Lout + l]
  m2d_outmap_plain(p.O, y, Lout, 1);
  p.O.cdiv = Lout;
  p.O.cdiv_inv = 1.f / (float)Lout;
  p.O.c_hi_stride = Cout * Lout;
  p.O.c_lo_stride = 1;
}

extern "C" {

// Replaces nn.Conv1d forward (+ fused bias / ReLU / LeakyReLU / residual add):
// phase3/archis/default.py:78-82,137-143,207-210,312-319,342-346.
// `out_mask` (optional, shape of y) multiplies the result by (mask>0 ? 1 : out_mask_slope):
// it is the d/d(dy) branch of backward-data's derivative (double backward of the GP).
int m2d_conv1d_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int L,
                   int Cout, int ks, int stride, int pad, int act, float slope, const float* residual,
                   const float* out_mask, float out_mask_slope, void* ws, size_t ws_bytes, void* stream) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || ks <= 0 || stride <= 0 || pad < 0 || L <= 0)
    M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_fwd: bad shape B=%d Cin=%d L=%d Cout=%d k=%d s=%d p=%d", B, Cin, L,
             Cout, ks, stride, pad);
  const int Lout = conv_out_len(L, ks, stride, pad);
  if (Lout <= 0) M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_fwd: empty output (L=%d k=%d s=%d p=%d)", L, ks, stride, pad);
  if (!fits_i32((long long)B * Cin * L) || !fits_i32((long long)B * Cout * Lout))
    M2D_FAIL(M2D_ERR_RANGE, "m2d_conv1d_fwd: tensor exceeds 2^31 elements");
  if (!residual && m2d_thin_applicable(Cin, Cout, ks, stride))
    return m2d_thin_fwd(x, w, bias, y, B, L, Cout, ks, stride, pad, Lout, act, slope, out_mask, out_mask_slope,
                        (hipStream_t)stream);
  M2dGemmParams p;
  fill_fwd(p, x, w, y, B, Cin, L, Cout, ks, stride, pad, Lout);
  p.O.bias = bias;
  p.O.bias_mode = bias ? 1 : 0;
  p.O.act = act;
  p.O.slope = slope;
  p.O.residual = residual;
  p.O.mask = out_mask;
  p.O.mask_slope = out_mask_slope;
  return m2d_gemm_launch(p, /*a_kfast=*/true, /*b_kfast=*/false, /*allow_split=*/true, ws, ws_bytes,
                         (hipStream_t)stream, "m2d_conv1d_fwd");
}

// Replaces the input-gradient half of convolution_backward (autograd of nn.Conv1d), the
// op the gradient penalty differentiates a second time (losses.py:40-44).
// `dy_mask` (optional, shape of dy): dy is read as dy * (mask>0 ? 1 : dy_mask_slope).
int m2d_conv1d_bwd_data(const float* dy, const float* w, float* dx, int B, int Cin, int L, int Cout,
                        int ks, int stride, int pad, const float* dy_mask, float dy_mask_slope, void* ws,
                        size_t ws_bytes, void* stream) {
  const int Lout = conv_out_len(L, ks, stride, pad);
  if (B <= 0 || Cin <= 0 || Cout <= 0 || Lout <= 0)
    M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_bwd_data: bad shape");
  if (!fits_i32((long long)B * Cin * L) || !fits_i32((long long)B * Cout * Lout) ||
      !fits_i32((long long)Cout * Cin * ks))
    M2D_FAIL(M2D_ERR_RANGE, "m2d_conv1d_bwd_data: tensor exceeds 2^31 elements");
  if (m2d_thin_applicable(Cin, Cout, ks, stride))
    return m2d_thin_bwd_data(dy, w, dx, B, L, Cout, ks, stride, pad, Lout, dy_mask, dy_mask_slope,
                             (hipStream_t)stream);
  M2dGemmParams p;
  memset(&p, 0, sizeof(p));
  if (Lout == 1 && pad == 0 && L == ks) {
    // full-length kernel (fconv / l6 / last encoder conv): dx[n,(ci,kk)] = sum_co dy[n,co] W[co,(ci,kk)]
    p.M = B;
    p.N = Cin * ks;
    p.K = Cout;
    p.phases = 1;
    m2d_operand_plain(p.A, dy, B, Cout, 1, (long long)B * Cout);
    p.A.mask = dy_mask;
    p.A.mask_slope = dy_mask_slope;
    m2d_operand_plain(p.B, w, p.N, 1, Cin * ks, (long long)Cout * Cin * ks);
    m2d_outmap_plain(p.O, dx, Cin * ks, 1);
    return m2d_gemm_launch(p, true, false, true, ws, ws_bytes, (hipStream_t)stream, "m2d_conv1d_bwd_data");
  }
  p.bwd_data = 1;
  p.phases = stride;
  p.ph_ks = ks;
  p.ph_cout = Cout;
  p.ph_pad = pad;
  p.ph_L = L;
  p.ph_batch = B;
  p.M = Cin;
  // widest phase: q in [qmin, (L-1+pad-r)/s]
  int nq_max = 0, k_max = 0;
  for (int r = 0; r < stride; ++r) {
    const int qmin = r >= pad ? 0 : (pad - r + stride - 1) / stride;
    const int top = L - 1 + pad - r;
    const int nq = top >= 0 ? (top / stride - qmin + 1) : 0;
    if (nq > nq_max) nq_max = nq;
    const int taps = r < ks ? (ks - r + stride - 1) / stride : 0;
    if (Cout * taps > k_max) k_max = Cout * taps;
  }
  p.N = B * nq_max;
  p.K = k_max;
  // A(m = ci, k = (co,t)) = W[co*Cin*ks + ci*ks + r + s*t]  (r added per phase on device)
  M2dOperand& a = p.A;
  a.base = w;
  a.nbytes = m2d_extent_bytes((long long)Cout * Cin * ks);
  a.nrows = Cin;
  a.rdiv = 1;
  a.rdiv_inv = 1.f;
  a.r_hi_stride = ks;
  a.kdiv = 1;  // taps(r), set on device
  a.kdiv_inv = 1.f;
  a.k_hi_stride = Cin * ks;
  a.k_lo_stride = stride;
  // B(k = (co,t), col = (n,q)) = dy[n*Cout*Lout + co*Lout + q - t], valid iff 0 <= q - t < Lout
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
  b.kdiv = 1;
  b.kdiv_inv = 1.f;
  b.k_hi_stride = Lout;
  b.k_lo_stride = -1;
  b.k_pos_mul = -1;
  b.lim = Lout;
  // out(m = ci, col = (n,q)) = dx[n*Cin*L + ci*L + s*q + r - pad]
  m2d_outmap_plain(p.O, dx, L, 1);
  p.O.c_hi_stride = Cin * L;
  p.O.c_lo_stride = stride;
  p.O.c_pos_mul = stride;
  p.O.c_lim = L;
  if (stride == 1) {
    // single phase: resolve the phase parameters here (r = 0, taps = ks, q in [pad, L-1+pad]) so the
    // launch is an ordinary GEMM and may be split along K (the TCN critic's small grids need it)
    p.bwd_data = 0;
    p.phases = 1;
    p.N = B * L;
    p.K = Cout * ks;
    a.kdiv = ks;
    a.kdiv_inv = 1.f / (float)ks;
    b.kdiv = ks;
    b.kdiv_inv = a.kdiv_inv;
    b.nrows = p.N;
    b.rdiv = L;
    b.rdiv_inv = 1.f / (float)L;
    b.r_off = pad;
    b.r_pos_off = pad;
    p.O.cdiv = L;
    p.O.cdiv_inv = b.rdiv_inv;
    p.O.c_off = 0;
    p.O.c_pos_off = 0;
    return m2d_gemm_launch(p, /*a_kfast=*/false, /*b_kfast=*/false, /*allow_split=*/true, ws, ws_bytes,
                           (hipStream_t)stream, "m2d_conv1d_bwd_data");
  }
  return m2d_gemm_launch(p, /*a_kfast=*/false, /*b_kfast=*/false, /*allow_split=*/false, ws, ws_bytes,
                         (hipStream_t)stream, "m2d_conv1d_bwd_data");
}

// Replaces the weight-gradient half of convolution_backward. K = B*Lout is the long
// dimension, so the launch is split-K with a deterministic slab reduction.
int m2d_conv1d_bwd_weight(const float* x, const float* dy, float* dw, int B, int Cin, int L, int Cout,
                          int ks, int stride, int pad, const float* dy_mask, float dy_mask_slope, void* ws,
                          size_t ws_bytes, void* stream) {
  const int Lout = conv_out_len(L, ks, stride, pad);
  if (B <= 0 || Cin <= 0 || Cout <= 0 || Lout <= 0)
    M2D_FAIL(M2D_ERR_ARG, "m2d_conv1d_bwd_weight: bad shape");
  if (!fits_i32((long long)B * Cin * L) || !fits_i32((long long)B * Cout * Lout))
    M2D_FAIL(M2D_ERR_RANGE, "m2d_conv1d_bwd_weight: tensor exceeds 2^31 elements");
  if (m2d_thin_applicable(Cin, Cout, ks, stride))
    return m2d_thin_bwd_weight(x, dy, dw, B, L, Cout, ks, stride, pad, Lout, dy_mask, dy_mask_slope, ws, ws_bytes,
                               (hipStream_t)stream);
  M2dGemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = Cout;
  p.N = Cin * ks;
  p.K = B * Lout;
  p.phases = 1;
  // A(m = co, k = (n,l)) = dy[n*Cout*Lout + co*Lout + l]
  M2dOperand& a = p.A;
  a.base = dy;
  a.nbytes = m2d_extent_bytes((long long)B * Cout * Lout);
  a.mask = dy_mask;
  a.mask_slope = dy_mask_slope;
  a.nrows = Cout;
  a.rdiv = 1;
  a.rdiv_inv = 1.f;
  a.r_hi_stride = Lout;
  a.kdiv = Lout;
  a.kdiv_inv = 1.f / (float)Lout;
  a.k_hi_stride = Cout * Lout;
  a.k_lo_stride = 1;
  // B(k = (n,l), col = (ci,kk)) = x[n*Cin*L + ci*L + l*s - pad + kk]
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
  b.kdiv = Lout;
  b.kdiv_inv = 1.f / (float)Lout;
  b.k_hi_stride = Cin * L;
  b.k_lo_stride = stride;
  b.k_pos_mul = stride;
  b.lim = L;
  m2d_outmap_plain(p.O, dw, Cin * ks, 1);
  return m2d_gemm_launch(p, /*a_kfast=*/true, /*b_kfast=*/false, /*allow_split=*/true, ws, ws_bytes,
                         (hipStream_t)stream, "m2d_conv1d_bwd_weight");
}

// which: 0 forward, 1 backward-data, 2 backward-weight
size_t m2d_conv1d_workspace_bytes(int which, int B, int Cin, int L, int Cout, int ks, int stride, int pad) {
  const int Lout = conv_out_len(L, ks, stride, pad);
  if (Lout <= 0) return 0;
  if (m2d_thin_applicable(Cin, Cout, ks, stride)) return which == 2 ? m2d_thin_bwd_weight_ws(B, Cout, ks, Lout) : 0;
  if (which == 0) return m2d_gemm_plan(Cout, B * Lout, Cin * ks, 1, true).ws_bytes;
  if (which == 1) {
    if (Lout == 1 && pad == 0 && L == ks) return m2d_gemm_plan(B, Cin * ks, Cout, 1, true).ws_bytes;
    if (stride == 1) return m2d_gemm_plan(Cin, B * L, Cout * ks, 1, true).ws_bytes;
    return 0;
  }
  return m2d_gemm_plan(Cout, Cin * ks, B * Lout, 1, true).ws_bytes;
}

// Dense row-major GEMMs behind nn.Linear (phase3/archis/default.py:153,161,176-177,256-257;
// phase1/archis/residual.py:11,19,35,41) and the GRU input projections:
//   mode 0 (NT): C[M,N] = A[M,K] * B[N,K]^T (+ bias[N], act)         y  = x W^T + b
//   mode 1 (NN): C[M,N] = A[M,K] * B[K,N]                              dx = dy W
//   mode 2 (TN): C[M,N] = A[K,M]^T * B[K,N]                            dW = dy^T x
// a_mask (shape of A) / out_mask (shape of C) fuse the activation derivative as in conv1d.
int m2d_gemm(int mode, const float* a, const float* b, const float* bias, float* c, int M, int N, int K,
             int act, float slope, const float* a_mask, float a_mask_slope, const float* out_mask,
             float out_mask_slope, void* ws, size_t ws_bytes, void* stream) {
  if (M < 0 || N < 0 || K < 0 || mode < 0 || mode > 2) M2D_FAIL(M2D_ERR_ARG, "m2d_gemm: bad arguments");
  if (!fits_i32((long long)M * K) || !fits_i32((long long)N * K) || !fits_i32((long long)M * N))
    M2D_FAIL(M2D_ERR_RANGE, "m2d_gemm: matrix exceeds 2^31 elements");
  M2dGemmParams p;
  memset(&p, 0, sizeof(p));
  p.M = M;
  p.N = N;
  p.K = K;
  p.phases = 1;
  bool akf = true, bkf = true;
  if (mode == 0) {
    m2d_operand_plain(p.A, a, M, K, 1, (long long)M * K);
    m2d_operand_plain(p.B, b, N, K, 1, (long long)N * K);
    akf = true;
    bkf = true;
  } else if (mode == 1) {
    m2d_operand_plain(p.A, a, M, K, 1, (long long)M * K);
    m2d_operand_plain(p.B, b, N, 1, N, (long long)N * K);
    akf = true;
    bkf = false;
  } else {
    m2d_operand_plain(p.A, a, M, 1, M, (long long)M * K);
    m2d_operand_plain(p.B, b, N, 1, N, (long long)N * K);
    akf = false;
    bkf = false;
  }
  p.A.mask = a_mask;
  p.A.mask_slope = a_mask_slope;
  m2d_outmap_plain(p.O, c, N, 1);
  p.O.bias = bias;
  p.O.bias_mode = bias ? 2 : 0;
  p.O.act = act;
  p.O.slope = slope;
  p.O.mask = out_mask;
  p.O.mask_slope = out_mask_slope;
  return m2d_gemm_launch(p, akf, bkf, true, ws, ws_bytes, (hipStream_t)stream, "m2d_gemm");
}

size_t m2d_gemm_workspace_bytes(int mode, int M, int N, int K) {
  (void)mode;
  return m2d_gemm_plan(M, N, K, 1, true).ws_bytes;
}

}  // extern "C"
