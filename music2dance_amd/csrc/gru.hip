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
