// GRU recurrence (one launch per time step) for nn.GRU(batch_first=True) as the
// reference's NoiseGen uses it (phase3/archis/default.py:349-355,
// phase2/archis/default.py:90-96). PyTorch gate order (r, z, n), h0 = 0:
//   gi = x W_ih^T + b_ih            (all T at once: one engine GEMM, not in this file)
//   gh = h_{t-1} W_hh^T + b_hh      (sequential: this file)
//   r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r * gh_n)
//   h_t = (1 - z) * n + z * h_{t-1}
//
// Each step is a skinny GEMM (batch x H x {H | 3H}) whose epilogue is the gate math, so
// a step is ONE kernel: a block owns 16 batch rows x 16 hidden units (all three gates),
// its 4 waves split K and combine through LDS (v_mfma_f32_16x16x4_f32, exact fp32).
// The step is latency-bound (<= 22 MFLOP), not roofline-bound; the grid is sized to put
// every (row-group, unit-slice) on its own CU.
#include "m2d_common.h"
#include <mutex>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GRU_UNROLL 16

__device__ __forceinline__ float gru_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

struct GruFwdArgs {
  const float* gi;      // (B, T, 3H), b_ih already added
  const float* w_hh_t;  // (H, 3H) = W_hh^T
  const float* b_hh;    // (3H)
  const int* lengths;   // optional (B): rows with t >= lengths[b] output 0
  float* out;           // (B, T, H)
  float* r_s;           // saved gates (B, T, H) each; may be NULL (inference)
  float* z_s;
  float* n_s;
  float* hn_s;          // W_hn h + b_hn
  int B, T, H, t;
};

__global__ void __launch_bounds__(256) m2d_gru_fwd_step_kernel(const GruFwdArgs a) {
  __shared__ float red[4][3][256];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int u0 = blockIdx.x * 16;
  const int b0 = blockIdx.y * 16;
  const int H = a.H, T = a.T, t = a.t;
  f32x4 acc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (t > 0) {
    const int arow = b0 + (lane & 15);
    const int bcol = u0 + (lane & 15);
    const bool rok = arow < a.B;
    const bool cok = bcol < H;
    const float* hp = a.out + ((size_t)arow * T + (t - 1)) * H;
    const int nsteps = (H + 3) / 4;
    // the step is pure latency: issue every operand load of a batch of GRU_UNROLL k-steps
    // before the first MFMA so the batch costs one memory round trip
    for (int s0 = wave; s0 < nsteps; s0 += 4 * GRU_UNROLL) {
      float av[GRU_UNROLL], bv[GRU_UNROLL][3];
#pragma unroll
      for (int i = 0; i < GRU_UNROLL; ++i) {
        const int k = 4 * (s0 + 4 * i) + (lane >> 4);
        const bool kok = (s0 + 4 * i) < nsteps && k < H;
        av[i] = (rok && kok) ? hp[k] : 0.f;
        const float* wrow = a.w_hh_t + (size_t)(kok ? k : 0) * 3 * H + (cok ? bcol : 0);
        const bool ok = kok && cok;
        bv[i][0] = ok ? wrow[0] : 0.f;
        bv[i][1] = ok ? wrow[H] : 0.f;
        bv[i][2] = ok ? wrow[2 * H] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < GRU_UNROLL; ++i) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i][1], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i][2], acc[2], 0, 0, 0);
      }
    }
  }
  // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][g][((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[g][r];
  __syncthreads();

  const int row = tid >> 4, col = tid & 15;
  const int b = b0 + row, u = u0 + col;
  if (b >= a.B || u >= H) return;
  float gh[3];
#pragma unroll
  for (int g = 0; g < 3; ++g)
    gh[g] = red[0][g][tid] + red[1][g][tid] + red[2][g][tid] + red[3][g][tid] + a.b_hh[g * H + u];
  const size_t bt = (size_t)b * T + t;
  const float* gi = a.gi + bt * 3 * H;
  const float hprev = t > 0 ? a.out[(bt - 1) * H + u] : 0.f;
  const float r = gru_sigmoid(gi[u] + gh[0]);
  const float z = gru_sigmoid(gi[H + u] + gh[1]);
  const float n = tanhf(gi[2 * H + u] + r * gh[2]);
  float h = (1.f - z) * n + z * hprev;
  if (a.lengths && t >= a.lengths[b]) h = 0.f;
  a.out[bt * H + u] = h;
  if (a.r_s) {
    a.r_s[bt * H + u] = r;
    a.z_s[bt * H + u] = z;
    a.n_s[bt * H + u] = n;
    a.hn_s[bt * H + u] = gh[2];
  }
}

struct GruBwdArgs {
  const float* dout;    // (B, T, H) gradient wrt the layer output
  const float* out;     // (B, T, H) forward outputs (h_t)
  const float* r_s;
  const float* z_s;
  const float* n_s;
  const float* hn_s;
  const float* w_hh;    // (3H, H)
  const int* lengths;   // optional
  float* dgi;           // (B, T, 3H): [dr_pre, dz_pre, dn_pre]
  float* dgh;           // (B, T, 3H): [dr_pre, dz_pre, dn_pre * r]
  float* dh_buf;        // (2, B, H) ping-pong of the total dL/dh_t
  int B, T, H, t;
};

__global__ void __launch_bounds__(256) m2d_gru_bwd_step_kernel(const GruBwdArgs a) {
  __shared__ float red[4][256];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int u0 = blockIdx.x * 16;
  const int b0 = blockIdx.y * 16;
  const int H = a.H, T = a.T, t = a.t;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool has_next = (t + 1) < T;
  if (has_next) {
    const int arow = b0 + (lane & 15);
    const int bcol = u0 + (lane & 15);
    const bool rok = arow < a.B;
    const bool cok = bcol < H;
    const float* dg = a.dgh + ((size_t)arow * T + (t + 1)) * 3 * H;
    const int K = 3 * H;
    const int nsteps = (K + 3) / 4;
    for (int s0 = wave; s0 < nsteps; s0 += 4 * GRU_UNROLL) {
      float av[GRU_UNROLL], bv[GRU_UNROLL];
#pragma unroll
      for (int i = 0; i < GRU_UNROLL; ++i) {
        const int k = 4 * (s0 + 4 * i) + (lane >> 4);
        const bool kok = (s0 + 4 * i) < nsteps && k < K;
        av[i] = (rok && kok) ? dg[k] : 0.f;
        bv[i] = (kok && cok) ? a.w_hh[(size_t)k * H + bcol] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < GRU_UNROLL; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];
  __syncthreads();

  const int row = tid >> 4, col = tid & 15;
  const int b = b0 + row, u = u0 + col;
  if (b >= a.B || u >= H) return;
  const size_t bt = (size_t)b * T + t;
  float dh = a.dout[bt * H + u];
  if (has_next) {
    const float rec = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    const float dh_next = a.dh_buf[((size_t)((t + 1) & 1) * a.B + b) * H + u];
    dh += dh_next * a.z_s[(bt + 1) * H + u] + rec;
  }
  const bool dead = a.lengths && t >= a.lengths[b];
  if (dead) dh = 0.f;
  const float r = a.r_s[bt * H + u], z = a.z_s[bt * H + u], n = a.n_s[bt * H + u], hn = a.hn_s[bt * H + u];
  const float hprev = t > 0 ? a.out[(bt - 1) * H + u] : 0.f;
  const float dn_pre = dh * (1.f - z) * (1.f - n * n);
  const float dz_pre = dh * (hprev - n) * z * (1.f - z);
  const float dr_pre = dn_pre * hn * r * (1.f - r);
  float* gi = a.dgi + bt * 3 * H;
  float* gh = a.dgh + bt * 3 * H;
  gi[u] = dr_pre;
  gi[H + u] = dz_pre;
  gi[2 * H + u] = dn_pre;
  gh[u] = dr_pre;
  gh[H + u] = dz_pre;
  gh[2 * H + u] = dn_pre * r;
  a.dh_buf[((size_t)(t & 1) * a.B + b) * H + u] = dh;
}

extern "C" {

// Runs the T sequential steps of one GRU layer. gi = x W_ih^T + b_ih must be precomputed
// (m2d_gemm mode 0). Saved gate tensors may all be NULL when no backward will follow.
// Replaces the recurrent half of nn.GRU forward (phase3/archis/default.py:352-355).
int m2d_gru_layer_fwd(const float* gi, const float* w_hh_t, const float* b_hh, const int* lengths,
                      float* out, float* r_s, float* z_s, float* n_s, float* hn_s, int B, int T, int H,
                      void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || T <= 0 || H <= 0) M2D_FAIL(M2D_ERR_ARG, "m2d_gru_layer_fwd: bad shape");
  if ((r_s == nullptr) != (z_s == nullptr) || (r_s == nullptr) != (n_s == nullptr) ||
      (r_s == nullptr) != (hn_s == nullptr))
    M2D_FAIL(M2D_ERR_ARG, "m2d_gru_layer_fwd: saved gate buffers must be all set or all NULL");
  GruFwdArgs a;
  a.gi = gi; a.w_hh_t = w_hh_t; a.b_hh = b_hh; a.lengths = lengths;
  a.out = out; a.r_s = r_s; a.z_s = z_s; a.n_s = n_s; a.hn_s = hn_s;
  a.B = B; a.T = T; a.H = H;
  dim3 grid(m2d_ceil_div(H, 16), m2d_ceil_div(B, 16));
  M2dProfScope prof(M2D_FAM_GRU, stream, 2.0 * B * 3.0 * H * H * (double)(T - 1), 0.0);
  for (int t = 0; t < T; ++t) {
    a.t = t;
    hipLaunchKernelGGL(m2d_gru_fwd_step_kernel, grid, dim3(256), 0, stream, a);
  }
  M2D_CHECK_LAUNCH("m2d_gru_fwd_step_kernel");
  return M2D_OK;
}

// Back-propagation through time for one layer: fills dgi and dgh (B, T, 3H each); the
// caller turns them into dW_ih, dW_hh, biases and dx with engine GEMMs.
// dh_buf: scratch of 2*B*H floats.
int m2d_gru_layer_bwd(const float* dout, const float* out, const float* r_s, const float* z_s,
                      const float* n_s, const float* hn_s, const float* w_hh, const int* lengths,
                      float* dgi, float* dgh, float* dh_buf, int B, int T, int H, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || T <= 0 || H <= 0) M2D_FAIL(M2D_ERR_ARG, "m2d_gru_layer_bwd: bad shape");
  GruBwdArgs a;
  a.dout = dout; a.out = out; a.r_s = r_s; a.z_s = z_s; a.n_s = n_s; a.hn_s = hn_s;
  a.w_hh = w_hh; a.lengths = lengths; a.dgi = dgi; a.dgh = dgh; a.dh_buf = dh_buf;
  a.B = B; a.T = T; a.H = H;
  dim3 grid(m2d_ceil_div(H, 16), m2d_ceil_div(B, 16));
  M2dProfScope prof(M2D_FAM_GRU, stream, 2.0 * B * 3.0 * H * H * (double)(T - 1), 0.0);
  for (int t = T - 1; t >= 0; --t) {
    a.t = t;
    hipLaunchKernelGGL(m2d_gru_bwd_step_kernel, grid, dim3(256), 0, stream, a);
  }
  M2D_CHECK_LAUNCH("m2d_gru_bwd_step_kernel");
  return M2D_OK;
}

}  // extern "C"

// =========================================================================================
// Stacked GRU on the (layer, t) diagonal: launch d runs layer l at time t = d - l for every
// layer at once (grid.z = layer), so a whole L-layer forward is T + L - 1 dependent launches
// instead of L * T. Layer l >= 1 needs h_{l-1}[t] (written by launch d - 1) and computes its
// input projection in the step: r and z accumulate W_ih x + W_hh h in one accumulator, the n
// gate keeps the two halves apart (n = tanh(gi_n + r * gh_n)). The backward runs the
// anti-diagonal e = (L-1-l) + (T-1-t) and folds dL/dh_l[t] = dgi_{l+1}[t] W_ih_{l+1} (upper
// layer) + dgh_l[t+1] W_hh_l (next step) into one K = 6H contraction.
#define GRU_MAX_LAYERS 4

struct GruStackFwdArgs {
  const float* gi0;                       // (B, T, 3H): layer-0 input projection incl. b_ih
  const float* w_ih_t[GRU_MAX_LAYERS];    // (H, 3H) = W_ih^T of layers >= 1 ([0] unused)
  const float* b_ih[GRU_MAX_LAYERS];      // (3H) of layers >= 1
  const float* w_hh_t[GRU_MAX_LAYERS];    // (H, 3H)
  const float* b_hh[GRU_MAX_LAYERS];
  float* out[GRU_MAX_LAYERS];             // (B, T, H)
  float* saved[GRU_MAX_LAYERS];           // (4, B, T, H): r, z, n, hn; NULL when not saving
  const int* lengths;
  int B, T, H, L, d;
};

template <int NB, int NW>
__device__ __forceinline__ void gru_mac(const float* arow, bool rok, const float* wt, int H, int ncol, int bcol,
                                        bool cok, int wave, int lane, f32x4 (&acc)[NB], const int (&col_of)[NB]) {
  // acc[j] += A[16 rows, K = H] * W^T[:, col_of[j] * H + bcol]; this wave takes k-steps wave, wave+NW, ...
  const int nsteps = (H + 3) / 4;
  for (int s0 = wave; s0 < nsteps; s0 += NW * GRU_UNROLL) {
    float av[GRU_UNROLL], bv[GRU_UNROLL][NB];
#pragma unroll
    for (int i = 0; i < GRU_UNROLL; ++i) {
      const int k = 4 * (s0 + NW * i) + (lane >> 4);
      const bool kok = (s0 + NW * i) < nsteps && k < H;
      av[i] = (rok && kok) ? arow[k] : 0.f;
      const float* wrow = wt + (size_t)(kok ? k : 0) * ncol + (cok ? bcol : 0);
      const bool ok = kok && cok;
#pragma unroll
      for (int j = 0; j < NB; ++j) bv[i][j] = ok ? wrow[col_of[j] * H] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < GRU_UNROLL; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i][j], acc[j], 0, 0, 0);
  }
}

// 8 waves split K: at H = 240 a wave multiplies 8 k-steps per contraction instead of 15 (the
// dependent MFMA chain and the operand burst per wave halve; the step is latency-bound)
#define GRU_FWD_NW 8
#define GRU_FWD_MAXB 2  // blocks of 16 k per wave: covers H <= 16 * 8 * 2 = 256
__global__ void __launch_bounds__(64 * GRU_FWD_NW) m2d_gru_stack_fwd_kernel(const GruStackFwdArgs a) {
  constexpr int NW = GRU_FWD_NW;
  __shared__ float red[NW][4][256];
  const int l = blockIdx.z;
  const int t = a.d - l;
  if (t < 0 || t >= a.T) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int H = a.H, T = a.T;
  // accumulators: 0 = r (input + hidden), 1 = z (input + hidden), 2 = gh_n, 3 = gi_n
  f32x4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int arow = b0 + (lane & 15);
  const int bcol = u0 + (lane & 15);
  const bool rok = arow < a.B, cok = bcol < H;
  // Both contractions (hidden: h_l[t-1] W_hh^T, input: h_{l-1}[t] W_ih^T) have K = H. When a
  // wave's share fits one batch (H <= 4 * NW * UN = 256) ALL operand loads of both are
  // issued before the first MFMA: one memory round trip per step instead of two.
  const bool use_h = t > 0, use_i = l > 0;
  if (H <= 16 * NW * GRU_FWD_MAXB) {
    // K = H in blocks of 16: wave w takes blocks w, w + NW; inside a block lane group q = lane >> 4 owns the four
    // consecutive k = 16 j + 4 q + m (MFMA k-step m multiplies the k's {16 j + m, + 4, + 8, + 12}): ONE 16-byte load
    // per lane and block - the rows were written by other CUs a step earlier and the step is bound by the number of
    // memory requests. The persistent kernel (below) uses the same assignment and order: bit-identical sums.
    const int nblk = (H + 15) / 16;
    const int q = lane >> 4;
    const bool vec = (H & 3) == 0;
    float ah[GRU_FWD_MAXB][4], ai[GRU_FWD_MAXB][4], bh[GRU_FWD_MAXB][4][3], bi[GRU_FWD_MAXB][4][3];
    const float* hrow = a.out[l] + ((size_t)arow * T + (use_h ? t - 1 : 0)) * H;
    const float* irow = a.out[use_i ? l - 1 : 0] + ((size_t)arow * T + t) * H;
    const float* wh = a.w_hh_t[l];
    const float* wi = a.w_ih_t[use_i ? l : 0];
#pragma unroll
    for (int i = 0; i < GRU_FWD_MAXB; ++i) {
      const int j = wave + NW * i;
      const int k0 = 16 * j + 4 * q;
      const bool jok = j < nblk;
      if (jok && vec && k0 + 3 < H) {
        const float4 vh = (use_h && rok) ? *reinterpret_cast<const float4*>(hrow + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 vi = (use_i && rok) ? *reinterpret_cast<const float4*>(irow + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
        ah[i][0] = vh.x; ah[i][1] = vh.y; ah[i][2] = vh.z; ah[i][3] = vh.w;
        ai[i][0] = vi.x; ai[i][1] = vi.y; ai[i][2] = vi.z; ai[i][3] = vi.w;
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const bool ok = jok && rok && k0 + m < H;
          ah[i][m] = (ok && use_h) ? hrow[k0 + m] : 0.f;
          ai[i][m] = (ok && use_i) ? irow[k0 + m] : 0.f;
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const bool ok = jok && cok && k0 + m < H;
        const size_t wo = (size_t)(ok ? k0 + m : 0) * 3 * H + (cok ? bcol : 0);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          bh[i][m][g] = (ok && use_h) ? wh[wo + g * H] : 0.f;
          bi[i][m][g] = (ok && use_i) ? wi[wo + g * H] : 0.f;
        }
      }
    }
    // input part first, hidden part second: the persistent kernel (below) can then multiply the layer
    // below's output while it still waits for its own layer's previous step; same order = same bits
    if (use_i) {
#pragma unroll
      for (int i = 0; i < GRU_FWD_MAXB; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[i][m], bi[i][m][0], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[i][m], bi[i][m][1], acc[1], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[i][m], bi[i][m][2], acc[3], 0, 0, 0);
        }
    }
    if (use_h) {
#pragma unroll
      for (int i = 0; i < GRU_FWD_MAXB; ++i)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i][m], bh[i][m][0], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i][m], bh[i][m][1], acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i][m], bh[i][m][2], acc[2], 0, 0, 0);
        }
    }
  } else {
    if (use_i) {
      f32x4 i3[3] = {acc[0], acc[1], acc[3]};
      const int cols[3] = {0, 1, 2};
      gru_mac<3, NW>(a.out[l - 1] + ((size_t)arow * T + t) * H, rok, a.w_ih_t[l], H, 3 * H, bcol, cok, wave, lane, i3, cols);
      acc[0] = i3[0]; acc[1] = i3[1]; acc[3] = i3[2];
    }
    if (use_h) {
      f32x4 h3[3] = {acc[0], acc[1], acc[2]};
      const int cols[3] = {0, 1, 2};
      gru_mac<3, NW>(a.out[l] + ((size_t)arow * T + (t - 1)) * H, rok, a.w_hh_t[l], H, 3 * H, bcol, cok, wave, lane, h3, cols);
      acc[0] = h3[0]; acc[1] = h3[1]; acc[2] = h3[2];
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][g][((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[g][r];
  __syncthreads();
  if (tid >= 256) return;
  const int row = tid >> 4, col = tid & 15;
  const int b = b0 + row, u = u0 + col;
  if (b >= a.B || u >= H) return;
  float s[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[w][g][tid];
    s[g] = v;
  }
  const size_t bt = (size_t)b * T + t;
  float gir, giz, gin;
  if (l == 0) {
    const float* gi = a.gi0 + bt * 3 * H;
    gir = gi[u]; giz = gi[H + u]; gin = gi[2 * H + u];
  } else {
    gir = a.b_ih[l][u]; giz = a.b_ih[l][H + u]; gin = s[3] + a.b_ih[l][2 * H + u];
  }
  const float* bh = a.b_hh[l];
  const float hn = s[2] + bh[2 * H + u];
  const float r = gru_sigmoid(gir + s[0] + bh[u]);
  const float z = gru_sigmoid(giz + s[1] + bh[H + u]);
  const float n = tanhf(gin + r * hn);
  const float hprev = t > 0 ? a.out[l][(bt - 1) * H + u] : 0.f;
  float h = (1.f - z) * n + z * hprev;
  if (a.lengths && t >= a.lengths[b]) h = 0.f;
  a.out[l][bt * H + u] = h;
  if (a.saved[l]) {
    const size_t plane = (size_t)a.B * T * H;
    float* sv = a.saved[l];
    sv[bt * H + u] = r;
    sv[plane + bt * H + u] = z;
    sv[2 * plane + bt * H + u] = n;
    sv[3 * plane + bt * H + u] = hn;
  }
}

// -----------------------------------------------------------------------------------------
// Persistent forward: ONE launch for the whole L-layer, T-step recurrence.
// Workgroup (hidden tile, batch tile, layer) keeps its weight slices - W_hh^T[:, 16 units x 3 gates]
// and, for layers >= 1, W_ih^T likewise: 2 x H x 48 floats = 92 KB at H = 240 - in LDS for all T
// steps (the per-step launches re-read them from L2 every step), and loops over t.
// Hand-off between workgroups: THE DATA IS THE FLAG. The launcher fills every layer's output with a sentinel bit
// pattern (0xFFFFFFFF, a NaN no arithmetic here produces); a producer publishes h_l[t] with write-through (`sc1`)
// dword stores and goes on; a consumer loads the 16 x H rows it needs with `sc1` loads (16 bytes per lane) and
// repeats the loads that still show the sentinel. Every dword is valid or sentinel on its own (4-byte stores are
// single-copy atomic), so nothing has to be ordered: no counters, no drain of the stores, no signalling lane, no
// acquire - one L2 round trip on the step's critical path instead of four (counter add, counter poll, barrier,
// operand load). The round-2/3 form (counter per (layer, batch tile), `sc1` poll, drain + barrier + add) took
// 7.5 us per step at H = 240; this one is bounded by store -> L2 -> load plus the 24 MFMAs and the gate math.
// Compute is m2d_gru_stack_fwd_kernel's: same k assignment, MFMA order and cross-wave sum (bit-identical results).
// All workgroups must be resident at once (checked by the launcher); every spin is bounded: on a timeout the wave
// raises `*error` (pinned host memory), every spinning wave sees it and leaves, so the kernel always terminates.
#define GRU_SENTINEL 0xFFFFFFFFu
#define GRU_CNT_STRIDE 64  // (backward) one counter per 256-byte line: pollers of different tiles do not share a line
struct GruPersistArgs {
  GruStackFwdArgs s;
  unsigned* error;     // host-visible word
  unsigned* mirror;    // its copy in device memory (what queued optimizer steps read: no PCIe round trip per workgroup)
  unsigned spin_limit;
};

typedef __attribute__((address_space(1))) unsigned gru_gu32;
typedef __attribute__((address_space(1))) float gru_gf32;
typedef unsigned int gru_u32x4 __attribute__((__vector_size__(16)));
typedef float gru_f32x4 __attribute__((__vector_size__(16)));

__device__ __forceinline__ float gru_ld_sc1(const float* p) {
  return __hip_atomic_load((gru_gf32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gru_st_sc1(float* p, float v) {
  __hip_atomic_store((gru_gf32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one lane: wait until *cnt >= want, or somebody raised the error word; false on timeout / error
__device__ __forceinline__ void gru_raise(unsigned* error, unsigned* mirror) {
  __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store((gru_gu32*)mirror, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool gru_wait_ge(unsigned* cnt, unsigned want, unsigned* error, unsigned* mirror, unsigned limit) {
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load((gru_gu32*)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    if ((spins & 63u) == 63u && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return false;
    if (spins >= limit) {
      gru_raise(error, mirror);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

// One lane's share (this wave's blocks) of a 16-row operand, through `sc1` loads; -> true when an element that is
// needed still shows the sentinel. rs: buffer descriptor over the (B, T, H) tensor; row_off: element offset of the
// lane's row at the step.
__device__ __forceinline__ bool gru_load_rows(__amdgpu_buffer_rsrc_t rs, const float* base, size_t row_off, bool rok, int H,
                                              int nblk, bool vec, int wave, int q, float (&av)[GRU_FWD_MAXB][4]) {
  constexpr int NW = GRU_FWD_NW;
  bool bad = false;
#pragma unroll
  for (int i = 0; i < GRU_FWD_MAXB; ++i) {
    const int j = wave + NW * i;
    const int k0 = 16 * j + 4 * q;
    if (j < nblk && rok) {
      if (vec && k0 + 3 < H) {
        // (whole-vector bit cast: see gru_bwd_contract)
        const gru_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((row_off + k0) << 2), 0, 16 /* sc1 */);
        bad = bad || v[0] == GRU_SENTINEL || v[1] == GRU_SENTINEL || v[2] == GRU_SENTINEL || v[3] == GRU_SENTINEL;
        const gru_f32x4 f = __builtin_bit_cast(gru_f32x4, v);
        av[i][0] = f[0]; av[i][1] = f[1]; av[i][2] = f[2]; av[i][3] = f[3];
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          float x = 0.f;
          if (k0 + m < H) {
            x = gru_ld_sc1(base + row_off + k0 + m);
            bad = bad || __builtin_bit_cast(unsigned, x) == GRU_SENTINEL;
          }
          av[i][m] = x;
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m) av[i][m] = 0.f;
    }
  }
  return bad;
}

// Both operands of a step (the layer below's h[t], the own layer's h[t-1]): all loads go out together, and the ones
// that came back with a sentinel are repeated until every lane of the wave has its data. -> false on timeout / error
// elsewhere.
__device__ __forceinline__ bool gru_spin_operands(bool use_i, __amdgpu_buffer_rsrc_t low_rs, const float* low, size_t low_off,
                                                  bool use_h, __amdgpu_buffer_rsrc_t own_rs, const float* own, size_t own_off,
                                                  bool rok, int H, int nblk, bool vec, int wave, int q, unsigned* error,
                                                  unsigned* mirror, unsigned limit, float (&ai)[GRU_FWD_MAXB][4], float (&ah)[GRU_FWD_MAXB][4],
                                                  unsigned& spin_acc) {
  bool need_i = use_i, need_h = use_h;
  for (unsigned spins = 0;; ++spins) {
    if (need_i) need_i = gru_load_rows(low_rs, low, low_off, rok, H, nblk, vec, wave, q, ai);
    if (need_h) need_h = gru_load_rows(own_rs, own, own_off, rok, H, nblk, vec, wave, q, ah);
    if (!__builtin_amdgcn_ballot_w64(need_i || need_h)) {
#ifdef GRU_COUNT_SPINS
      spin_acc += spins;
#endif
      return true;  // wave-uniform: every lane has its operands
    }
    if ((spins & 63u) == 63u && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return false;
    if (spins >= limit) {
      gru_raise(error, mirror);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

__global__ void __launch_bounds__(64 * GRU_FWD_NW) m2d_gru_persist_fwd_kernel(const GruPersistArgs pa) {
  constexpr int NW = GRU_FWD_NW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const GruStackFwdArgs& a = pa.s;
  const int H = a.H, T = a.T;
  float (*red)[4][256] = reinterpret_cast<float (*)[4][256]>(lds);
  const int l = blockIdx.z, bt = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = blockIdx.x * 16, b0 = bt * 16;
  const bool use_i = l > 0;
  const int arow = b0 + (lane & 15);
  const bool rok = arow < a.B;
  const int nblk = (H + 15) / 16;
  const int q = lane >> 4;
  const bool vec = (H & 3) == 0;
  // The weight slices live in REGISTERS for all T steps: a lane's MFMA B operands are W^T[k][gate g, unit u0 + (lane & 15)]
  // for its 2 blocks x 4 k's x 3 gates - 24 floats per contraction, 48 with the input part. (Rounds 2-3 kept the slices
  // in LDS, 92 KB at H = 240: three LDS reads in front of every MFMA triple on the step's critical path, and a footprint
  // that kept the workgroup off any CU still holding two GEMM workgroups. LDS is now the 32 KB cross-wave sum only.)
  float bwh[GRU_FWD_MAXB][4][3], bwi[GRU_FWD_MAXB][4][3];
  {
    const int ucol = u0 + (lane & 15);
#pragma unroll
    for (int i = 0; i < GRU_FWD_MAXB; ++i) {
      const int j = wave + NW * i;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int k = 16 * j + 4 * q + m;
        const bool ok = j < nblk && k < H && ucol < H;
        const size_t src = (size_t)(ok ? k : 0) * 3 * H + (ok ? ucol : 0);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          bwh[i][m][g] = ok ? a.w_hh_t[l][src + (size_t)g * H] : 0.f;
          bwi[i][m][g] = (ok && use_i) ? a.w_ih_t[l][src + (size_t)g * H] : 0.f;
        }
      }
    }
  }
  const unsigned obytes = (unsigned)((size_t)a.B * T * H * sizeof(float));  // < 2^31: checked by the launcher
  const __amdgpu_buffer_rsrc_t own_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.out[l], (short)0, (int)obytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t low_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.out[use_i ? l - 1 : l], (short)0, (int)obytes, 0x00020000);
  const int row = (tid & 255) >> 4, col = tid & 15;
  const int b = b0 + row, u = u0 + col;
  const bool owner = tid < 256 && b < a.B && u < H;
  float hprev = 0.f;  // this thread's own previous output (owner threads)
  float bhr = 0.f, bhz = 0.f, bhn = 0.f, bir = 0.f, biz = 0.f, bin = 0.f;
  if (owner) {
    bhr = a.b_hh[l][u]; bhz = a.b_hh[l][H + u]; bhn = a.b_hh[l][2 * H + u];
    if (use_i) { bir = a.b_ih[l][u]; biz = a.b_ih[l][H + u]; bin = a.b_ih[l][2 * H + u]; }
  }

  unsigned spin_acc = 0;  // (GRU_COUNT_SPINS builds: repeated operand loads of this wave)
  // layer 0: the input projection (written before the launch, (B, T, 3H): every step touches lines nobody has read
  // yet - an HBM round trip) is fetched ONE STEP AHEAD, so the gate math never waits for it
  float nir = bir, niz = biz, nin = bin;
  if (owner && l == 0) {
    const float* gi = a.gi0 + ((size_t)b * T) * 3 * H;
    nir = gi[u]; niz = gi[H + u]; nin = gi[2 * H + u];
  }
  for (int t = 0; t < T; ++t) {
    const bool use_h = t > 0;
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float gir = nir, giz = niz, gin = nin;
    if (owner && l == 0 && t + 1 < T) {
      const float* gi = a.gi0 + ((size_t)b * T + t + 1) * 3 * H;
      nir = gi[u]; niz = gi[H + u]; nin = gi[2 * H + u];
    }
    // ---- operands: step t of the layer below (usually there already) and step t-1 of every hidden tile of this
    //      layer (the critical dependency)
    float ai[GRU_FWD_MAXB][4], ah[GRU_FWD_MAXB][4];
    if (!gru_spin_operands(use_i, low_rs, a.out[use_i ? l - 1 : l], ((size_t)arow * T + t) * H, use_h, own_rs, a.out[l],
                           ((size_t)arow * T + (use_h ? t - 1 : 0)) * H, rok, H, nblk, vec, wave, q, pa.error, pa.mirror,
                           pa.spin_limit, ai, ah, spin_acc))
      return;  // timeout or error elsewhere (waves that have left are not counted by the barriers)
    if (use_i) {
#pragma unroll
      for (int i = 0; i < GRU_FWD_MAXB; ++i) {
        const int j = wave + NW * i;
        if (j < nblk) {
          const int k0 = 16 * j + 4 * q;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[i][m], bwi[i][m][0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[i][m], bwi[i][m][1], acc[1], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[i][m], bwi[i][m][2], acc[3], 0, 0, 0);
          }
        }
      }
    }
    if (use_h) {
#pragma unroll
      for (int i = 0; i < GRU_FWD_MAXB; ++i) {
        const int j = wave + NW * i;
        if (j < nblk) {
          const int k0 = 16 * j + 4 * q;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i][m], bwh[i][m][0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i][m], bwh[i][m][1], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i][m], bwh[i][m][2], acc[2], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][g][((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[g][r];
    __syncthreads();
    if (owner) {
      float sg[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][g][tid];
        sg[g] = v;
      }
      const size_t bt_ = (size_t)b * T + t;
      if (l != 0) gin = sg[3] + bin;
      const float hn = sg[2] + bhn;
      const float r = gru_sigmoid(gir + sg[0] + bhr);
      const float z = gru_sigmoid(giz + sg[1] + bhz);
      const float n = tanhf(gin + r * hn);
      float h = (1.f - z) * n + z * hprev;
      if (a.lengths && t >= a.lengths[b]) h = 0.f;
      hprev = h;
      gru_st_sc1(a.out[l] + bt_ * H + u, h);  // published: consumers spin on the value itself
      if (a.saved[l]) {
        const size_t plane = (size_t)a.B * T * H;
        float* sv = a.saved[l];
        sv[bt_ * H + u] = r;
        sv[plane + bt_ * H + u] = z;
        sv[2 * plane + bt_ * H + u] = n;
        sv[3 * plane + bt_ * H + u] = hn;
      }
    }
    __syncthreads();  // `red` is free for the next step
  }
#ifdef GRU_COUNT_SPINS
  if (lane == 0) { atomicAdd(pa.error + 1, spin_acc); atomicAdd(pa.error + 2, (unsigned)T); }
#endif
}

struct GruStackBwdArgs {
  const float* dout;                      // (B, T, H): gradient w.r.t. the TOP layer's output
  const float* out[GRU_MAX_LAYERS];
  const float* saved[GRU_MAX_LAYERS];     // (4, B, T, H)
  const float* w_hh[GRU_MAX_LAYERS];      // (3H, H)
  const float* w_ih[GRU_MAX_LAYERS];      // (3H, H) of layers >= 1 ([0] unused)
  float* dgi[GRU_MAX_LAYERS];             // (B, T, 3H)
  float* dgh[GRU_MAX_LAYERS];             // (B, T, 3H)
  float* dh_buf[GRU_MAX_LAYERS];          // (2, B, H)
  const int* lengths;
  int B, T, H, L, e;
};

// acc += A[16 rows, K] * W[K, H][:, bcol]. K is walked in blocks of 16: wave w takes blocks w, w + NW, ...; inside a
// block lane group q = lane >> 4 owns the four consecutive k = 16 j + 4 q + m, m = 0..3 (MFMA k-step m multiplies the
// k's {16 j + m, + 4, + 8, + 12}): ONE 16-byte load per lane and block instead of four scattered dwords - the A rows
// are gradients written by other CUs a step earlier, and the step is bound by the number of memory requests. The
// persistent kernel below uses the same assignment and order (bit-identical sums).
template <int NW>
__device__ __forceinline__ void gru_mac1(const float* arow, bool rok, const float* w, int K, int H, int bcol, bool cok,
                                         int wave, int lane, f32x4& acc) {
  const int nblk = (K + 15) / 16;
  const int q = lane >> 4;
  const bool vec = (K & 3) == 0;
  for (int j = wave; j < nblk; j += NW) {
    const int k0 = 16 * j + 4 * q;
    float av[4], bv[4];
    if (vec && k0 + 3 < K) {
      const float4 v = rok ? *reinterpret_cast<const float4*>(arow + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w;
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m) av[m] = (rok && k0 + m < K) ? arow[k0 + m] : 0.f;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) bv[m] = (cok && k0 + m < K) ? w[(size_t)(k0 + m) * H + bcol] : 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[m], acc, 0, 0, 0);
  }
}

#define GRU_BWD_NW 8
__global__ void __launch_bounds__(64 * GRU_BWD_NW) m2d_gru_stack_bwd_kernel(const GruStackBwdArgs a) {
  constexpr int NW = GRU_BWD_NW;
  __shared__ float red[NW][256];
  const int l = a.L - 1 - (int)blockIdx.z;
  const int t = a.T - 1 - (a.e - (int)blockIdx.z);
  if (t < 0 || t >= a.T) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
  const int H = a.H, T = a.T;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int arow = b0 + (lane & 15);
  const int bcol = u0 + (lane & 15);
  const bool rok = arow < a.B, cok = bcol < H;
  const bool has_next = (t + 1) < T;
  const bool has_up = (l + 1) < a.L;
  // upper-layer part first, next-step part second: the persistent kernel (below) can then multiply the layer above's
  // gradient while it still waits for its own layer's next step; same order = same bits
  if (has_up) gru_mac1<NW>(a.dgi[l + 1] + ((size_t)arow * T + t) * 3 * H, rok, a.w_ih[l + 1], 3 * H, H, bcol, cok, wave, lane, acc);
  if (has_next) gru_mac1<NW>(a.dgh[l] + ((size_t)arow * T + (t + 1)) * 3 * H, rok, a.w_hh[l], 3 * H, H, bcol, cok, wave, lane, acc);
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];
  __syncthreads();
  if (tid >= 256) return;
  const int row = tid >> 4, col = tid & 15;
  const int b = b0 + row, u = u0 + col;
  if (b >= a.B || u >= H) return;
  const size_t bt = (size_t)b * T + t;
  const size_t plane = (size_t)a.B * T * H;
  const float* sv = a.saved[l];
  float dh = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) dh += red[w][tid];
  if (!has_up) dh += a.dout[bt * H + u];
  if (has_next) dh += a.dh_buf[l][((size_t)((t + 1) & 1) * a.B + b) * H + u] * sv[plane + (bt + 1) * H + u];
  if (a.lengths && t >= a.lengths[b]) dh = 0.f;
  const float r = sv[bt * H + u], z = sv[plane + bt * H + u], n = sv[2 * plane + bt * H + u], hn = sv[3 * plane + bt * H + u];
  const float hprev = t > 0 ? a.out[l][(bt - 1) * H + u] : 0.f;
  const float dn_pre = dh * (1.f - z) * (1.f - n * n);
  const float dz_pre = dh * (hprev - n) * z * (1.f - z);
  const float dr_pre = dn_pre * hn * r * (1.f - r);
  float* gi = a.dgi[l] + bt * 3 * H;
  float* gh = a.dgh[l] + bt * 3 * H;
  gi[u] = dr_pre; gi[H + u] = dz_pre; gi[2 * H + u] = dn_pre;
  gh[u] = dr_pre; gh[H + u] = dz_pre; gh[2 * H + u] = dn_pre * r;
  a.dh_buf[l][((size_t)(t & 1) * a.B + b) * H + u] = dh;
}

// -----------------------------------------------------------------------------------------
// Persistent backward: ONE launch for the whole L-layer BPTT (round 3; same hand-off as the forward).
// Workgroup (hidden tile, batch tile, layer) keeps W_hh_l[:, 16 units] and W_ih_{l+1}[:, 16 units] (2 x 3H x 16
// floats = 92 KB at H = 240) in LDS and walks t = T-1 .. 0:
//   upper part   dgi_{l+1}[t] (16 rows x 3H, written by the layer above's workgroups in this launch: `sc1` loads
//                after its counter shows step t done) times W_ih_{l+1};
//   next part    dgh_l[t+1] (own layer, all hidden tiles: the critical dependency) times W_hh_l;
//   gates        dh = sum (+ dout for the top layer) + dh_{t+1} * z_{t+1} (the owner thread's own register),
//                gate derivatives -> dgi_l[t], dgh_l[t] with `sc1` stores; drain, barrier, ONE lane signals.
// K-step assignment, MFMA order and the cross-wave sum are those of m2d_gru_stack_bwd_kernel: bit-identical.
struct GruPersistBwdArgs {
  GruStackBwdArgs s;
  unsigned* counters;  // [L][nbt] x GRU_CNT_STRIDE, zeroed before the launch
  unsigned* error;
  unsigned* mirror;
  unsigned spin_limit;
};

// The persistent kernel's contraction: gru_mac1's block assignment, the A rows through `sc1` loads (16 bytes per
// lane when K is a multiple of 4: raw buffer load, aux = sc1), W from the workgroup's LDS slice [K][16]. All loads of
// the wave's (at most 6 at H = 240) blocks are issued before the first MFMA.
#define GRU_BWD_MAXB 6  // blocks per wave: covers K = 3H <= 16 * 8 * 6 = 768 (H <= 256)
template <int NW>
__device__ __forceinline__ void gru_bwd_contract(const float* base, size_t row_off, bool rok,
                                                 const float (&bw)[GRU_BWD_MAXB][4], int K, int wave, int lane, bool vec,
                                                 __amdgpu_buffer_rsrc_t rs, f32x4& acc) {
  const int nblk = (K + 15) / 16;
  const int q = lane >> 4;
  float av[GRU_BWD_MAXB][4];
#pragma unroll
  for (int i = 0; i < GRU_BWD_MAXB; ++i) {
    const int j = wave + NW * i;
    const int k0 = 16 * j + 4 * q;
    if (j < nblk && vec && k0 + 3 < K) {
      // (rows past the batch read through offset 0x80000000: the descriptor's range check returns zeros)
      const unsigned off = rok ? (unsigned)((row_off + k0) << 2) : 0x80000000u;
      // (whole-vector bit cast: element-wise __builtin_bit_cast(float, v[i]) on the builtin's result is narrowed to
      // ONE dword load by hipcc 7.2 - every element then reads as the first)
      const gru_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16 /* sc1 */);
      const gru_f32x4 f = __builtin_bit_cast(gru_f32x4, v);
      av[i][0] = f[0]; av[i][1] = f[1]; av[i][2] = f[2]; av[i][3] = f[3];
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m) av[i][m] = (j < nblk && rok && k0 + m < K) ? gru_ld_sc1(base + row_off + k0 + m) : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < GRU_BWD_MAXB; ++i) {
    const int j = wave + NW * i;
    if (j < nblk) {
      const int k0 = 16 * j + 4 * q;
#pragma unroll
      for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][m], bw[i][m], acc, 0, 0, 0);
    }
  }
}

__global__ void __launch_bounds__(64 * GRU_BWD_NW) m2d_gru_persist_bwd_kernel(const GruPersistBwdArgs pa) {
  constexpr int NW = GRU_BWD_NW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const GruStackBwdArgs& a = pa.s;
  const int H = a.H, T = a.T, K = 3 * a.H;
  const bool vec = (K & 3) == 0;
  const unsigned gbytes = (unsigned)((size_t)a.B * T * K * sizeof(float));  // < 2^31: checked by the launcher
  const __amdgpu_buffer_rsrc_t gh_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dgh[blockIdx.z], (short)0, (int)gbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gi_rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.dgi[(int)blockIdx.z + 1 < a.L ? blockIdx.z + 1 : blockIdx.z], (short)0, (int)gbytes, 0x00020000);
  float (*red)[256] = reinterpret_cast<float (*)[256]>(lds);
  __shared__ int go;
  const int l = blockIdx.z, bt = blockIdx.y;
  const int nth = gridDim.x, nbt = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = blockIdx.x * 16, b0 = bt * 16;
  const bool has_up = (l + 1) < a.L;
  // the weight slices (W_hh_l[k][u0 + c], W_ih_{l+1}[k][u0 + c]: a lane's MFMA B operands for its 6 blocks x 4 k's) live
  // in registers for all T steps, as in the forward kernel: no LDS read in front of the MFMAs, LDS = the 8 KB sum
  float bwhh[GRU_BWD_MAXB][4], bwih[GRU_BWD_MAXB][4];
  {
    const int ucol = u0 + (lane & 15);
    const int nblk = (K + 15) / 16;
#pragma unroll
    for (int i = 0; i < GRU_BWD_MAXB; ++i) {
      const int j = wave + NW * i;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int k = 16 * j + 4 * (lane >> 4) + m;
        const bool ok = j < nblk && k < K && ucol < H;
        bwhh[i][m] = ok ? a.w_hh[l][(size_t)k * H + ucol] : 0.f;
        bwih[i][m] = (ok && has_up) ? a.w_ih[l + 1][(size_t)k * H + ucol] : 0.f;
      }
    }
  }
  const int arow = b0 + (lane & 15);
  const bool rok = arow < a.B;
  const int row = (tid & 255) >> 4, col = tid & 15;
  const int b = b0 + row, u = u0 + col;
  const bool owner = tid < 256 && b < a.B && u < H;
  const size_t plane = (size_t)a.B * T * H;
  const float* sv = a.saved[l];
  float dh_next = 0.f;  // this thread's dL/dh_{t+1} (owner threads)
  unsigned* my_cnt = pa.counters + ((size_t)l * nbt + bt) * GRU_CNT_STRIDE;
  unsigned* up_cnt = pa.counters + ((size_t)(has_up ? l + 1 : l) * nbt + bt) * GRU_CNT_STRIDE;

  for (int t = T - 1; t >= 0; --t) {
    const bool has_next = (t + 1) < T;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    // values written before the launch: fetched under the waits
    float r = 0.f, z = 0.f, n = 0.f, hn = 0.f, hprev = 0.f, znext = 0.f, dtop = 0.f;
    if (owner) {
      const size_t bt_ = (size_t)b * T + t;
      r = sv[bt_ * H + u]; z = sv[plane + bt_ * H + u]; n = sv[2 * plane + bt_ * H + u]; hn = sv[3 * plane + bt_ * H + u];
      hprev = t > 0 ? a.out[l][(bt_ - 1) * H + u] : 0.f;
      znext = has_next ? sv[plane + (bt_ + 1) * H + u] : 0.f;
      dtop = has_up ? 0.f : a.dout[bt_ * H + u];
    }
    // ---- upper part: step t of the layer above (it runs one step ahead of this layer)
    if (has_up) {
      if (tid == 0) go = gru_wait_ge(up_cnt, (unsigned)nth * (unsigned)(T - t), pa.error, pa.mirror, pa.spin_limit) ? 1 : 0;
      __syncthreads();
      if (!go) return;
      gru_bwd_contract<NW>(a.dgi[l + 1], ((size_t)arow * T + t) * K, rok, bwih, K, wave, lane, vec, gi_rs, acc);
    }
    // ---- next part: step t+1 of every hidden tile of this layer
    if (has_next) {
      if (nth > 1) {
        __syncthreads();  // everybody is past the previous read of `go`
        if (tid == 0) go = gru_wait_ge(my_cnt, (unsigned)nth * (unsigned)(T - 1 - t), pa.error, pa.mirror, pa.spin_limit) ? 1 : 0;
        __syncthreads();
        if (!go) return;
      }
      gru_bwd_contract<NW>(a.dgh[l], ((size_t)arow * T + (t + 1)) * K, rok, bwhh, K, wave, lane, vec, gh_rs, acc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][((lane >> 4) * 4 + q) * 16 + (lane & 15)] = acc[q];
    __syncthreads();
    if (owner) {
      const size_t bt_ = (size_t)b * T + t;
      float dh = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) dh += red[w][tid];
      if (!has_up) dh += dtop;
      if (has_next) dh += dh_next * znext;
      if (a.lengths && t >= a.lengths[b]) dh = 0.f;
      const float dn_pre = dh * (1.f - z) * (1.f - n * n);
      const float dz_pre = dh * (hprev - n) * z * (1.f - z);
      const float dr_pre = dn_pre * hn * r * (1.f - r);
      float* gi = a.dgi[l] + bt_ * K;
      float* gh = a.dgh[l] + bt_ * K;
      gru_st_sc1(gi + u, dr_pre); gru_st_sc1(gi + H + u, dz_pre); gru_st_sc1(gi + 2 * H + u, dn_pre);
      gru_st_sc1(gh + u, dr_pre); gru_st_sc1(gh + H + u, dz_pre); gru_st_sc1(gh + 2 * H + u, dn_pre * r);
      dh_next = dh;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add((gru_gu32*)my_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- persistent-launch state (per device): counters, host-visible error word, limits -------------
struct GruPersistState {
  unsigned* error_host = nullptr;  // hipHostMalloc (mapped): the kernel raises it, the host reads it
  unsigned* error_dev = nullptr;
  unsigned* mirror = nullptr;      // device-memory copy of the word, raised with it (m2d_async_fault_word)
  unsigned spin_limit = 0;
  int cus = 0;
  int max_lds = 0;
  bool usable = false;
};
static GruPersistState g_gru_ps[16];
static bool g_gru_ps_init[16];
static std::mutex g_gru_mu;

static GruPersistState* gru_persist_state_peek() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(g_gru_mu);
  return g_gru_ps_init[dev] ? &g_gru_ps[dev] : nullptr;
}

static GruPersistState* gru_persist_state() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(g_gru_mu);
  GruPersistState& ps = g_gru_ps[dev];
  if (!g_gru_ps_init[dev]) {
    g_gru_ps_init[dev] = true;
    const char* e = getenv("M2D_PERSISTENT_GRU");
    if (e && e[0] == '0') return &ps;  // usable stays false
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return &ps;
    ps.cus = prop.multiProcessorCount;
    ps.max_lds = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (hipHostMalloc((void**)&ps.error_host, 64, hipHostMallocMapped) != hipSuccess) return &ps;
    *ps.error_host = 0u;
    if (hipHostGetDevicePointer((void**)&ps.error_dev, ps.error_host, 0) != hipSuccess) return &ps;
    if (hipMalloc((void**)&ps.mirror, 64) != hipSuccess || hipMemset(ps.mirror, 0, 64) != hipSuccess) return &ps;
    // ~0.4 us per poll with s_sleep(4): 2^21 polls ~ 1 s before a workgroup gives up
    ps.spin_limit = 1u << 21;
    if (hipFuncSetAttribute((const void*)m2d_gru_persist_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            140 * 1024) != hipSuccess)
      return &ps;
    if (hipFuncSetAttribute((const void*)m2d_gru_persist_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            140 * 1024) != hipSuccess)
      return &ps;
    ps.usable = true;
  }
  return ps.usable ? &ps : nullptr;
}

static size_t gru_persist_lds(int H) { (void)H; return (size_t)GRU_FWD_NW * 4 * 256 * sizeof(float); }  // the cross-wave sum only
static size_t gru_persist_bwd_lds(int H) { (void)H; return (size_t)GRU_BWD_NW * 256 * sizeof(float); }  // the cross-wave sum only

// every workgroup must be resident at once. ONE per CU: a workgroup is 8 waves of 131-174 VGPRs (the weight slices
// live in registers), i.e. two waves per SIMD, and a second workgroup's waves do not fit beside them - the LDS
// footprint (32 KB / 8 KB) no longer says so by itself.
static bool gru_persist_ok(dim3 grid, int H, int T, bool backward = false) {
  if (H > 256 || T < 4) return false;
  GruPersistState* ps = gru_persist_state();
  if (!ps) return false;
  const size_t lds = backward ? gru_persist_bwd_lds(H) : gru_persist_lds(H);
  if ((int)lds > ps->max_lds) return false;
  const long long blocks = (long long)grid.x * grid.y * grid.z;
  // leave a margin: other streams' kernels (the noise GRU) need a place to run too
  return blocks <= (long long)ps->cus * 3 / 4;
}

extern "C" {

// scratch words m2d_gru_stack_fwd needs for its persistent form (multiple of 4: the memset stays 16-byte sized)
int m2d_gru_stack_counters(int B, int L) { return L * m2d_ceil_div(B, 16) * GRU_CNT_STRIDE; }

// L-layer GRU forward on the (layer, t) diagonal. Pointer arrays have L entries; entry 0 of
// w_ih_t / b_ih is ignored (layer 0's projection gi0 is precomputed by m2d_gemm).
// saved[l]: (4, B, T, H) or all NULL.
int m2d_gru_stack_fwd(const float* gi0, const float* const* w_ih_t, const float* const* b_ih,
                      const float* const* w_hh_t, const float* const* b_hh, float* const* out, float* const* saved,
                      const int* lengths, int B, int T, int H, int L, unsigned* counters, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || T <= 0 || H <= 0 || L <= 0 || L > GRU_MAX_LAYERS) M2D_FAIL(M2D_ERR_ARG, "m2d_gru_stack_fwd: bad shape");
  GruStackFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.gi0 = gi0; a.lengths = lengths;
  for (int l = 0; l < L; ++l) {
    a.w_ih_t[l] = w_ih_t[l]; a.b_ih[l] = b_ih[l]; a.w_hh_t[l] = w_hh_t[l]; a.b_hh[l] = b_hh[l];
    a.out[l] = out[l]; a.saved[l] = saved ? saved[l] : nullptr;
  }
  a.B = B; a.T = T; a.H = H; a.L = L;
  dim3 grid(m2d_ceil_div(H, 16), m2d_ceil_div(B, 16), L);
  M2dProfScope prof(M2D_FAM_GRU, stream, 2.0 * B * 3.0 * H * H * (double)T * (2 * L - 1), 0.0, "gru_stack_fwd", B, T, H);
  // `counters` non-NULL (the backward's scratch size; the forward only takes it as the request): run the recurrence
  // as ONE persistent launch when every workgroup fits on the chip at once
  if (counters && gru_persist_ok(grid, H, T) && (long long)B * T * H * 4 < 0x7fffffffLL) {
    GruPersistState* ps = gru_persist_state();
    if (ps) {
      // every output starts as the sentinel the consumers spin on (one memset when the layers' outputs are one
      // allocation, as kernels.py makes them)
      const size_t obytes = sizeof(float) * (size_t)B * T * H;
      bool contiguous = true;
      for (int l = 1; l < L; ++l) contiguous = contiguous && (out[l] == out[l - 1] + (size_t)B * T * H);
      bool ok = true;
      if (contiguous) ok = hipMemsetAsync(out[0], 0xFF, obytes * L, stream) == hipSuccess;
      else
        for (int l = 0; l < L; ++l) ok = ok && hipMemsetAsync(out[l], 0xFF, obytes, stream) == hipSuccess;
      if (ok) {
        GruPersistArgs pa;
        pa.s = a;
        pa.error = ps->error_dev;
      pa.mirror = ps->mirror;
        pa.spin_limit = ps->spin_limit;
        hipLaunchKernelGGL(m2d_gru_persist_fwd_kernel, grid, dim3(64 * GRU_FWD_NW), gru_persist_lds(H), stream, pa);
        M2D_CHECK_LAUNCH("m2d_gru_persist_fwd_kernel");
        return M2D_OK;
      }
    }
  }
  for (int d = 0; d < T + L - 1; ++d) {
    a.d = d;
    hipLaunchKernelGGL(m2d_gru_stack_fwd_kernel, grid, dim3(64 * GRU_FWD_NW), 0, stream, a);
  }
  M2D_CHECK_LAUNCH("m2d_gru_stack_fwd_kernel");
  return M2D_OK;
}

// 1 when a persistent GRU launch timed out since the last call (and clears the flag): the outputs
// of that call are invalid. 0 otherwise. Does not synchronise: call it after a stream sync.
int m2d_gru_persist_error(void) {
  GruPersistState* ps = gru_persist_state_peek();
  if (!ps || !ps->error_host) return 0;
#ifdef GRU_COUNT_SPINS
  {
    volatile unsigned* w = (volatile unsigned*)ps->error_host;
    if (w[2]) fprintf(stderr, "[gru spins] %u retries over %u wave-steps = %.2f per step\n", w[1], w[2], (double)w[1] / w[2]);
    w[1] = 0; w[2] = 0;
  }
#endif
  const unsigned e = *(volatile unsigned*)ps->error_host;
  if (e) {
    *(volatile unsigned*)ps->error_host = 0u;
    if (ps->mirror) (void)hipMemset(ps->mirror, 0, 4);  // (the caller has synchronised: recovery path only)
  }
  return e ? 1 : 0;
}

// The same word without clearing it (recovery protocol, engine.py: peek -> device synchronise, so that every queued
// optimizer step has seen the word and skipped itself -> m2d_gru_persist_error() to clear -> go on with step launches)
int m2d_gru_persist_peek(void) {
  GruPersistState* ps = gru_persist_state_peek();
  if (!ps || !ps->error_host) return 0;
  return *(volatile unsigned*)ps->error_host ? 1 : 0;
}

// Address of the word's DEVICE-MEMORY copy (a timed-out launch raises both), or NULL when the persistent form is not
// in use: what m2d_adam_multi takes as `skip` - an optimizer step queued behind a recurrence that gave up voids
// itself. (Not the mapped host word: every workgroup of the step reads it, and 30 k PCIe round trips per step cost
// 1.9 ms of a 12.2 ms C3 body - measured, round 4.)
void* m2d_async_fault_word(void) {
  GruPersistState* ps = gru_persist_state();
  return ps ? (void*)ps->mirror : nullptr;
}

// dst[0] = 1.0f when the word is raised, else 0.0f (on `stream`): the form in which a data-parallel gradient exchange
// carries this rank's state to its peers - one more float in the last bucket, MAX-like after the sum / average.
__global__ void m2d_fault_fetch_kernel(const unsigned* word, float* dst) {
  dst[0] = (word && __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1.f : 0.f;
}
int m2d_fault_fetch(float* dst, void* stream) {
  if (!dst) M2D_FAIL(M2D_ERR_ARG, "m2d_fault_fetch: NULL destination");
  GruPersistState* ps = gru_persist_state();
  hipLaunchKernelGGL(m2d_fault_fetch_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ps ? ps->mirror : nullptr, dst);
  M2D_CHECK_LAUNCH("m2d_fault_fetch");
  return M2D_OK;
}

// Test hook: raise the word as a timed-out launch would.
int m2d_gru_persist_raise(void) {
  GruPersistState* ps = gru_persist_state();
  if (!ps || !ps->error_host) return M2D_ERR_ARG;
  *(volatile unsigned*)ps->error_host = 1u;
  const unsigned one = 1u;
  if (ps->mirror && hipMemcpy(ps->mirror, &one, 4, hipMemcpyHostToDevice) != hipSuccess) return M2D_ERR_HIP;
  return M2D_OK;
}

// BPTT for the whole stack on the anti-diagonal. dgi[l], dgh[l]: (B, T, 3H); dh_buf[l]: 2*B*H floats.
int m2d_gru_stack_bwd(const float* dout, const float* const* out, const float* const* saved,
                      const float* const* w_hh, const float* const* w_ih, float* const* dgi, float* const* dgh,
                      float* const* dh_buf, const int* lengths, int B, int T, int H, int L, unsigned* counters,
                      void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || T <= 0 || H <= 0 || L <= 0 || L > GRU_MAX_LAYERS) M2D_FAIL(M2D_ERR_ARG, "m2d_gru_stack_bwd: bad shape");
  GruStackBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.dout = dout; a.lengths = lengths;
  for (int l = 0; l < L; ++l) {
    a.out[l] = out[l]; a.saved[l] = saved[l]; a.w_hh[l] = w_hh[l]; a.w_ih[l] = w_ih[l];
    a.dgi[l] = dgi[l]; a.dgh[l] = dgh[l]; a.dh_buf[l] = dh_buf[l];
  }
  a.B = B; a.T = T; a.H = H; a.L = L;
  dim3 grid(m2d_ceil_div(H, 16), m2d_ceil_div(B, 16), L);
  M2dProfScope prof(M2D_FAM_GRU, stream, 2.0 * B * 3.0 * H * H * (double)T * (2 * L - 1), 0.0, "gru_stack_bwd", B, T, H);
  // `counters` (optional, m2d_gru_stack_counters(B, L) unsigneds owned by this call): the whole BPTT as ONE
  // persistent launch when every workgroup fits on the chip at once
  if (counters && gru_persist_ok(grid, H, T, true) && (long long)B * T * 3 * H * 4 < 0x7fffffffLL) {
    GruPersistState* ps = gru_persist_state();
    if (ps && hipMemsetAsync(counters, 0, sizeof(unsigned) * (size_t)m2d_gru_stack_counters(B, L), stream) == hipSuccess) {
      GruPersistBwdArgs pa;
      pa.s = a;
      pa.counters = counters;
      pa.error = ps->error_dev;
      pa.mirror = ps->mirror;
      pa.spin_limit = ps->spin_limit;
      hipLaunchKernelGGL(m2d_gru_persist_bwd_kernel, grid, dim3(64 * GRU_BWD_NW), gru_persist_bwd_lds(H), stream, pa);
      M2D_CHECK_LAUNCH("m2d_gru_persist_bwd_kernel");
      return M2D_OK;
    }
  }
  for (int e = 0; e < T + L - 1; ++e) {
    a.e = e;
    hipLaunchKernelGGL(m2d_gru_stack_bwd_kernel, grid, dim3(64 * GRU_BWD_NW), 0, stream, a);
  }
  M2D_CHECK_LAUNCH("m2d_gru_stack_bwd_kernel");
  return M2D_OK;
}

}  // extern "C"
