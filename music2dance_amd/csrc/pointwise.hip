// HBM-bound elementwise / reduction kernels of the WGAN-GP step:
//   * gradient-penalty interpolate and per-sample grad-norm penalty (losses.py:13-25,47-60)
//   * L1 loss and total-variation loss with their gradients (phase3/train.py:226,
//     losses.py:76-82)
//   * MaxPool1d(2,2) and Upsample(x2, linear, align_corners=False) of the U-Net encoder
//     (phase3/archis/default.py:235-245)
// Reductions are two-stage and deterministic: per-block fp64 partials, then one block.
#include "m2d_common.h"

// ---------------------------------------------------------------- block reduce helper
__device__ __forceinline__ double block_sum_256(double v, double* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) sh[t] += sh[t + s];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(256) m2d_finish_sum_kernel(const double* partial, int n, double scale,
                                                             float* out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) out[0] = (float)(s * scale);
}

// ---------------------------------------------------------------- GP interpolate
// out[b,i] = alpha[b] * real[b,i] + (1 - alpha[b]) * fake[b,i], three separately rounded
// fp32 operations exactly like the reference expression (losses.py:20).
__global__ void __launch_bounds__(256) m2d_gp_interpolate_kernel(const float* real, const float* fake,
                                                                 const float* alpha, float* out, int B,
                                                                 int n) {
  const int b = blockIdx.y;
  const float al = alpha[b];
  const float om = __fsub_rn(1.0f, al);
  const size_t base = (size_t)b * n;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float t1 = __fmul_rn(al, real[base + i]);
    const float t2 = __fmul_rn(om, fake[base + i]);
    out[base + i] = __fadd_rn(t1, t2);
  }
}

// ---------------------------------------------------------------- GP norm penalty
// norms[b] = sqrt(sum_i g[b,i]^2 + eps)   (eps = 1e-12 for GP, 0 for LP; losses.py:47-54)
// Stage 1: GP_SPLIT blocks per sample (a 76 800-element audio gradient is 307 KB: one block per
// sample would leave 3/4 of the CUs idle and be latency-bound), float4 loads, fp32 partials
// flushed to fp64 every 16 terms; partial[b][s] is written without atomics (deterministic).
#define GP_SPLIT 16
__global__ void __launch_bounds__(256) m2d_gp_norm_partial_kernel(const float* g, double* partial, int n) {
  __shared__ double sh[256];
  const int b = blockIdx.y, sp = blockIdx.x;
  const float* row = g + (size_t)b * n;
  const int per = ((n + GP_SPLIT - 1) / GP_SPLIT + 3) & ~3;
  const int i0 = sp * per;
  int i1 = i0 + per;
  if (i1 > n) i1 = n;
  double acc = 0.0;
  const bool vec = (((size_t)row) & 15) == 0 && (n % 4) == 0;
  if (vec) {
    for (int i = i0 + 4 * threadIdx.x; i + 3 < i1; i += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(row + i);
      acc += (double)(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
    }
    const int tail = i0 + ((i1 - i0) & ~3);
    for (int i = tail + threadIdx.x; i < i1; i += 256) acc += (double)(row[i] * row[i]);
  } else {
    float s = 0.f;
    int cnt = 0;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
      const float v = row[i];
      s += v * v;
      if (++cnt == 16) {
        acc += (double)s;
        s = 0.f;
        cnt = 0;
      }
    }
    acc += (double)s;
  }
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partial[(size_t)b * GP_SPLIT + sp] = acc;
}

// Stage 2 (one block): norms[b] = sqrt(sum_s partial[b][s] + eps); penalty = mean_b (norm_b - 1)^2
// (lp == 0) or mean_b max(0, norm_b - 1)^2 (lp != 0)
__global__ void __launch_bounds__(256) m2d_gp_penalty_kernel(const double* partial, float* norms, int B, int lp,
                                                             float eps, float* out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) {
    double q = 0.0;
    for (int k = 0; k < GP_SPLIT; ++k) q += partial[(size_t)b * GP_SPLIT + k];
    const float nb = sqrtf((float)q + eps);
    norms[b] = nb;
    float d = nb - 1.0f;
    if (lp && d < 0.f) d = 0.f;
    s += (double)(d * d);
  }
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) out[0] = (float)(s / (double)B);
}

// dg[b,i] = gout * 2 (norm_b - 1) / (norm_b * B) * g[b,i]   (LP: 0 where norm_b <= 1)
__global__ void __launch_bounds__(256) m2d_gp_penalty_bwd_kernel(const float* g, const float* norms,
                                                                 const float* gout, float* dg, int B, int n,
                                                                 int lp) {
  const int b = blockIdx.y;
  const float nb = norms[b];
  float d = nb - 1.0f;
  if (lp && d < 0.f) d = 0.f;
  const float coef = (d == 0.f) ? 0.f : gout[0] * 2.0f * d / (nb * (float)B);
  const size_t base = (size_t)b * n;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dg[base + i] = coef * g[base + i];
}

// ---------------------------------------------------------------- L1 / TV
// partial sums of |a[i] - b[i]|
__global__ void __launch_bounds__(256) m2d_absdiff_partial_kernel(const float* a, const float* b, size_t n,
                                                                  double* partial) {
  __shared__ double sh[256];
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    s += (double)fabsf(a[i] - b[i]);
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// d/da mean|a - b| = gout * sign(a - b) / n
__global__ void __launch_bounds__(256) m2d_absdiff_bwd_kernel(const float* a, const float* b, const float* gout,
                                                              float* da, size_t n) {
  const float sc = gout[0] / (float)n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float d = a[i] - b[i];
    da[i] = d > 0.f ? sc : (d < 0.f ? -sc : 0.f);
  }
}

// total variation over the time axis of a (B, C, T) tensor stored with strides
// (sb, sc, st) in elements: mean over B*C*(T-1) of |x[t+1] - x[t]|.
__global__ void __launch_bounds__(256) m2d_tv_partial_kernel(const float* x, int B, int C, int T, long sb,
                                                             long sc, long st, double* partial) {
  __shared__ double sh[256];
  const size_t total = (size_t)B * C * (T - 1);
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i % (size_t)(T - 1));
    const size_t bc = i / (size_t)(T - 1);
    const int c = (int)(bc % (size_t)C);
    const int b = (int)(bc / (size_t)C);
    const float* p = x + b * sb + c * sc + t * st;
    s += (double)fabsf(p[st] - p[0]);
  }
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void __launch_bounds__(256) m2d_tv_bwd_kernel(const float* x, const float* gout, float* dx, int B,
                                                         int C, int T, long sb, long sc, long st) {
  const size_t total = (size_t)B * C * T;
  const float scl = gout[0] / (float)((size_t)B * C * (T - 1));
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i % (size_t)T);
    const size_t bc = i / (size_t)T;
    const int c = (int)(bc % (size_t)C);
    const int b = (int)(bc / (size_t)C);
    const float* p = x + b * sb + c * sc + t * st;
    float g = 0.f;
    if (t > 0) {
      const float d = p[0] - p[-st];
      g += d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    }
    if (t + 1 < T) {
      const float d = p[st] - p[0];
      g -= d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    }
    dx[b * sb + c * sc + t * st] = g * scl;
  }
}

// jerkiness (losses.py:85-89): third finite difference along time of a (B, C, T) tensor given by element
// strides, squared, summed over channels, mean over (B, T-3): sum_{b,c,t} d^2 / (B * (T - 3)).
__global__ void __launch_bounds__(256) m2d_jerk_partial_kernel(const float* x, int B, int C, int T, long sb,
                                                               long sc, long st, double* partial) {
  __shared__ double sh[256];
  const size_t total = (size_t)B * C * (T - 3);
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i % (size_t)(T - 3));
    const size_t bc = i / (size_t)(T - 3);
    const int c = (int)(bc % (size_t)C);
    const int b = (int)(bc / (size_t)C);
    const float* p = x + b * sb + c * sc + t * st;
    // the reference's expression, left to right in fp32: x[t+3] - 3 x[t+2] + 3 x[t+1] - x[t]
    // (separately rounded products and sums, as torch evaluates it: no contraction into FMAs)
    const float d = __fsub_rn(__fadd_rn(__fsub_rn(p[3 * st], __fmul_rn(3.f, p[2 * st])), __fmul_rn(3.f, p[st])), p[0]);
    s += (double)__fmul_rn(d, d);
  }
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// y[r, c] = x[r, c] * scale[c] + shift[c] (MinMaxScaler.transform / inverse_transform of the reference's datasets,
// utils.py:26-31,79-85: two separately rounded fp32 operations, like numpy's X *= scale; X += min)
__global__ void __launch_bounds__(256) m2d_affine_cols_kernel(const float* x, const float* scale, const float* shift,
                                                              float* y, size_t rows, int cols) {
  const size_t total = rows * (size_t)cols;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % (size_t)cols);
    y[i] = __fadd_rn(__fmul_rn(x[i], scale[c]), shift[c]);
  }
}

// ---------------------------------------------------------------- multi-tensor Adam (+ packed conv-weight images)
// One launch steps up to M2D_ADAM_BATCH tensors (their records travel in the kernel arguments, like torch's
// multi_tensor_apply). 28 algorithmic bytes per element (read p, g, m, v; write p, m, v) + 8 per element of a conv weight
// whose packed images are refreshed on the way. Round 5 (the round-4 kernel moved 309 MB in 121-141 us = 0.28-0.32 of
// HBM peak: one dword per lane and access, and the two packed images written as 4-byte stores a whole row apart):
//   plain tensors    a workgroup owns 4 096 consecutive elements, 16 bytes per lane and access;
//   packed weights   a workgroup owns a tile of 8 output channels x 256 (ci, kk) positions: the four streams are read and
//                    written as 16-byte runs along (ci, kk), the updated tile is kept in LDS and leaves once more as the
//                    forward image (ci, kk, co) - 8 consecutive co per run - and as the backward image (co, kk, ci) - the
//                    tile's ~256 / ks consecutive ci per run.
// The arithmetic and its order are unchanged (bit-equal to torch.optim.Adam(foreach=False), tests/test_optim.py).
#define M2D_ADAM_BATCH 48
#define M2D_ADAM_CHUNK 4096
#define M2D_ADAM_TCO 8
#define M2D_ADAM_TJ 256
#define M2D_ADAM_VEC 1   // reserved bit 0: every pointer 16-byte aligned
#define M2D_ADAM_TILE 2  // reserved bit 1: tile mode (packed images, cin * ks a multiple of 4, aligned)
struct M2dAdamBatch {
  M2dAdamItem it[M2D_ADAM_BATCH];
  int first_block[M2D_ADAM_BATCH + 1];
  int n;
};

__device__ __forceinline__ void m2d_adam_one(float g, float& m, float& v, float& p, float omb1, float beta2, float omb2,
                                             float bc2_sqrt, float eps, float lr_over_bc1) {
  m = m + omb1 * (g - m);
  v = beta2 * v + omb2 * g * g;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - lr_over_bc1 * (m / denom);
}

__global__ void __launch_bounds__(256) m2d_adam_multi_kernel(const M2dAdamBatch b, float lr_over_bc1, float beta1, float beta2,
                                                             float eps, float bc2_sqrt, const float* skip) {
  // (any non-zero BIT pattern counts: the word may be an unsigned flag, e.g. m2d_async_fault_word(); read past the
  // caches - it can be raised by a kernel that ran just before)
  if (skip && __hip_atomic_load((const unsigned*)skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  __shared__ float P[M2D_ADAM_TCO][M2D_ADAM_TJ + 4];
  int k = 0;
  while (k + 1 < b.n && (int)blockIdx.x >= b.first_block[k + 1]) ++k;   // wave-uniform
  const M2dAdamItem& t = b.it[k];
  const int lb = (int)blockIdx.x - b.first_block[k];
  const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  const int tid = threadIdx.x;
  if (t.reserved & M2D_ADAM_TILE) {
    const int J = t.cin * t.ks;
    const int tiles_j = (J + M2D_ADAM_TJ - 1) / M2D_ADAM_TJ;
    const int tco = lb / tiles_j, tj = lb - tco * tiles_j;
    const int co0 = tco * M2D_ADAM_TCO, j0 = tj * M2D_ADAM_TJ;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int sidx = tid + 256 * e, r = sidx >> 6, c4 = sidx & 63;
      const int co = co0 + r, j = j0 + 4 * c4;
      if (co < t.cout && j < J) {   // (J % 4 == 0: the four lanes are inside together)
        const long long i = (long long)co * J + j;
        const float4 g = *reinterpret_cast<const float4*>(t.grad + i);
        float4 m = *reinterpret_cast<const float4*>(t.exp_avg + i), v = *reinterpret_cast<const float4*>(t.exp_avg_sq + i);
        float4 p = *reinterpret_cast<const float4*>(t.param + i);
        m2d_adam_one(g.x, m.x, v.x, p.x, omb1, beta2, omb2, bc2_sqrt, eps, lr_over_bc1);
        m2d_adam_one(g.y, m.y, v.y, p.y, omb1, beta2, omb2, bc2_sqrt, eps, lr_over_bc1);
        m2d_adam_one(g.z, m.z, v.z, p.z, omb1, beta2, omb2, bc2_sqrt, eps, lr_over_bc1);
        m2d_adam_one(g.w, m.w, v.w, p.w, omb1, beta2, omb2, bc2_sqrt, eps, lr_over_bc1);
        *reinterpret_cast<float4*>(t.exp_avg + i) = m;
        *reinterpret_cast<float4*>(t.exp_avg_sq + i) = v;
        *reinterpret_cast<float4*>(t.param + i) = p;
        P[r][4 * c4] = p.x; P[r][4 * c4 + 1] = p.y; P[r][4 * c4 + 2] = p.z; P[r][4 * c4 + 3] = p.w;
      }
    }
    __syncthreads();
    if (t.pack_fwd) {  // (ci, kk, co): element (co, j) at j * cout + co - 8 consecutive co per run
      const int co_l = tid & (M2D_ADAM_TCO - 1), co = co0 + co_l;
#pragma unroll
      for (int e = 0; e < M2D_ADAM_TJ / 32; ++e) {
        const int jj = (tid >> 3) + 32 * e, j = j0 + jj;
        if (co < t.cout && j < J) t.pack_fwd[(long long)j * t.cout + co] = P[co_l][jj];
      }
    }
    if (t.pack_bwd) {  // (co, kk, ci): element (co, j = ci * ks + kk) at (co * ks + kk) * cin + ci - consecutive ci per run
      const int r = tid >> 5, u = tid & 31, co = co0 + r;
      const int ci_first = j0 / t.ks;
      const int j_end = (j0 + M2D_ADAM_TJ < J) ? j0 + M2D_ADAM_TJ : J;
      const int nci = (j_end - 1) / t.ks - ci_first + 1;
      if (co < t.cout)
        for (int idx = u; idx < nci * t.ks; idx += 32) {
          const int kk = idx / nci, ci = ci_first + (idx - kk * nci);
          const int j = ci * t.ks + kk;
          if (j >= j0 && j < j_end) t.pack_bwd[((long long)co * t.ks + kk) * t.cin + ci] = P[r][j - j0];
        }
    }
    return;
  }
  const long long base = (long long)lb * M2D_ADAM_CHUNK;
  if (t.reserved & M2D_ADAM_VEC) {
#pragma unroll
    for (int e = 0; e < M2D_ADAM_CHUNK / 1024; ++e) {
      const long long i = base + 4LL * (tid + 256 * e);
      if (i + 3 < t.numel) {
        const float4 g = *reinterpret_cast<const float4*>(t.grad + i);
        float4 m = *reinterpret_cast<const float4*>(t.exp_avg + i), v = *reinterpret_cast<const float4*>(t.exp_avg_sq + i);
        float4 p = *reinterpret_cast<const float4*>(t.param + i);
        m2d_adam_one(g.x, m.x, v.x, p.x, omb1, beta2, omb2, bc2_sqrt, eps, lr_over_bc1);
        m2d_adam_one(g.y, m.y, v.y, p.y, omb1, beta2, omb2, bc2_sqrt, eps, lr_over_bc1);
        m2d_adam_one(g.z, m.z, v.z, p.z, omb1, beta2, omb2, bc2_sqrt, eps, lr_over_bc1);
        m2d_adam_one(g.w, m.w, v.w, p.w, omb1, beta2, omb2, bc2_sqrt, eps, lr_over_bc1);
        *reinterpret_cast<float4*>(t.exp_avg + i) = m;
        *reinterpret_cast<float4*>(t.exp_avg_sq + i) = v;
        *reinterpret_cast<float4*>(t.param + i) = p;
      } else {
        for (long long q = i; q < t.numel && q < i + 4; ++q) {
          float m = t.exp_avg[q], v = t.exp_avg_sq[q], p = t.param[q];
          m2d_adam_one(t.grad[q], m, v, p, omb1, beta2, omb2, bc2_sqrt, eps, lr_over_bc1);
          t.exp_avg[q] = m; t.exp_avg_sq[q] = v; t.param[q] = p;
        }
      }
    }
    return;
  }
  // unaligned tensors / packed weights whose rows are not multiples of 4 (the pose critic's 69-channel first layer):
  // one element per lane and access, the packed images as scattered stores
#pragma unroll 4
  for (int e = 0; e < M2D_ADAM_CHUNK / 256; ++e) {
    const long long i = base + e * 256 + tid;
    if (i >= t.numel) break;
    float m = t.exp_avg[i], v = t.exp_avg_sq[i], p = t.param[i];
    m2d_adam_one(t.grad[i], m, v, p, omb1, beta2, omb2, bc2_sqrt, eps, lr_over_bc1);
    t.exp_avg[i] = m;
    t.exp_avg_sq[i] = v;
    t.param[i] = p;
    if (t.pack_fwd || t.pack_bwd) {  // (co, ci, kk) -> (ci, kk, co) and (co, kk, ci)
      const int kk = (int)(i % t.ks);
      const long long r = i / t.ks;
      const int ci = (int)(r % t.cin), co = (int)(r / t.cin);
      if (t.pack_fwd) t.pack_fwd[((long long)ci * t.ks + kk) * t.cout + co] = p;
      if (t.pack_bwd) t.pack_bwd[((long long)co * t.ks + kk) * t.cin + ci] = p;
    }
  }
}

// ---------------------------------------------------------------- tanh heads (`activ: tanh`)
// The 'tanh' switch of the encoders / critics (phase3/archis/default.py:75-76,102-103,134-135,309-310,339-340) on
// the (N, code) head tensors, with the two derivatives the gradient penalty's double backward needs:
//   y = tanh(x);   gx = gy * (1 - y^2);   d(gx)/d(gy) = g * (1 - y^2),  d(gx)/d(y) = -2 * y * g * gy
__global__ void __launch_bounds__(256) m2d_tanh_fwd_kernel(const float* x, float* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = tanhf(x[i]);
}
__global__ void __launch_bounds__(256) m2d_tanh_bwd_kernel(const float* gy, const float* y, float* gx, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = y[i];
    gx[i] = gy[i] * (1.f - v * v);
  }
}
__global__ void __launch_bounds__(256) m2d_tanh_bwd_bwd_kernel(const float* g, const float* gy, const float* y,
                                                               float* g_y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    g_y[i] = -2.f * y[i] * g[i] * gy[i];
}

// ---------------------------------------------------------------- critic-iteration input pack
// The critic iteration's three pose batches in ONE channels-first buffer (3B, C, T) - [interpolated | real | fake] -
// from the loader's real poses (B, T, C) and the generator's rows (B*T, C) (phase3/train.py:196-199 permute +
// contiguous, losses.py:13-25 interpolate): a (32 frames x C joints) tile is read as one contiguous run, staged in
// LDS, and leaves as 128-byte runs along time per joint row. The interpolation is the reference's expression,
// three separately rounded fp32 operations.
#define M2D_PACK_TT 32
__global__ void __launch_bounds__(256) m2d_pose_pack3_kernel(const float* __restrict__ real,
                                                             const float* __restrict__ fake,
                                                             const float* __restrict__ alpha, float* __restrict__ out,
                                                             int B, int T, int C) {
  extern __shared__ float sh[];  // [2][TT][C + 1]
  const int b = blockIdx.y, t0 = blockIdx.x * M2D_PACK_TT;
  const int nt = min(M2D_PACK_TT, T - t0);
  const int ld = C + 1;
  float* sr = sh;
  float* sf = sh + M2D_PACK_TT * ld;
  const size_t base = ((size_t)b * T + t0) * C;
  for (int i = threadIdx.x; i < nt * C; i += 256) {
    const int t = i / C, c = i - t * C;
    sr[t * ld + c] = real[base + i];
    sf[t * ld + c] = fake[base + i];
  }
  __syncthreads();
  const float a = alpha[b];
  const float na = __fsub_rn(1.f, a);
  const size_t plane = (size_t)C * T;
  float* oi = out + (size_t)b * plane + t0;
  float* orl = out + ((size_t)B + b) * plane + t0;
  float* of = out + ((size_t)2 * B + b) * plane + t0;
  for (int i = threadIdx.x; i < C * M2D_PACK_TT; i += 256) {
    const int c = i / M2D_PACK_TT, t = i - c * M2D_PACK_TT;
    if (t < nt) {
      const float r = sr[t * ld + c], f = sf[t * ld + c];
      oi[(size_t)c * T + t] = __fadd_rn(__fmul_rn(a, r), __fmul_rn(na, f));
      orl[(size_t)c * T + t] = r;
      of[(size_t)c * T + t] = f;
    }
  }
}

// loss scalars of a critic iteration from the (3B,) scores [interpolated | real | fake] and the penalty term(s):
// out[0] = E[D(fake)] - E[D(real)] + gamma * gp, out[1] = gp, out[2] = E[D(fake)] - E[D(real)]
// (phase3/train.py:204-212, phase2/train.py:146-153). One block; fp64 accumulation.
__global__ void __launch_bounds__(256) m2d_wgan_critic_loss_kernel(const float* scores, int B, const float* pen0,
                                                                   const float* pen1, float gamma, float* out) {
  __shared__ double sh[256];
  double sr = 0.0, sf = 0.0;
  for (int i = threadIdx.x; i < B; i += 256) {
    sr += (double)scores[B + i];
    sf += (double)scores[2 * B + i];
  }
  sr = block_sum_256(sr, sh);
  sf = block_sum_256(sf, sh);
  if (threadIdx.x == 0) {
    const float er = (float)(sr / B), ef = (float)(sf / B);
    const float gp = pen1 ? __fadd_rn(pen0[0], pen1[0]) : pen0[0];
    const float w = __fsub_rn(ef, er);
    out[0] = __fadd_rn(w, __fmul_rn(gamma, gp));
    out[1] = gp;
    out[2] = w;
  }
}

// ---------------------------------------------------------------- pool / upsample
// MaxPool1d(2,2): y[r, j] = max(x[r, 2j], x[r, 2j+1]), rows = B*C, Lout = L/2
__global__ void __launch_bounds__(256) m2d_maxpool2_fwd_kernel(const float* x, float* y, size_t rows, int L,
                                                               int Lout) {
  const size_t total = rows * Lout;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / Lout;
    const int j = (int)(i - r * Lout);
    const float a = x[r * L + 2 * j], b = x[r * L + 2 * j + 1];
    y[i] = a > b ? a : b;  // ties take the first element's value (same number either way)
  }
}

// L even: rows do not matter - y[i] = max(x[2i], x[2i+1]) over the flat tensor, four outputs (32 B in, 16 B out) per
// thread and step, no index arithmetic (the row form above does a 64-bit division per output: 5.3 -> 6+ TB/s)
// (vps > 0: the input's samples - vps 8-float vectors each - are xpitch floats apart: a channel block of a wider buffer)
__global__ void __launch_bounds__(256) m2d_maxpool2_fwd_flat_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                    size_t nvec, size_t vps, long long xpitch) {
#pragma unroll 2
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 256) {
    size_t xo = 8 * v;
    if (vps) {
      const size_t smp = v / vps;
      xo = smp * (size_t)xpitch + 8 * (v - smp * vps);
    }
    const float4 a = *reinterpret_cast<const float4*>(x + xo);
    const float4 b = *reinterpret_cast<const float4*>(x + xo + 4);
    float4 o;
    o.x = a.x > a.y ? a.x : a.y;
    o.y = a.z > a.w ? a.z : a.w;
    o.z = b.x > b.y ? b.x : b.y;
    o.w = b.z > b.w ? b.z : b.w;
    *reinterpret_cast<float4*>(y + 4 * v) = o;
  }
}

// the same for the gradient: dx[2i], dx[2i+1] from (x[2i], x[2i+1], dy[i])
__global__ void __launch_bounds__(256) m2d_maxpool2_bwd_flat_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                    float* __restrict__ dx, size_t nvec) {
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 256) {
    const float4 a = *reinterpret_cast<const float4*>(x + 8 * v);
    const float4 b = *reinterpret_cast<const float4*>(x + 8 * v + 4);
    const float4 g = *reinterpret_cast<const float4*>(dy + 4 * v);
    const bool f0 = a.x >= a.y || a.x != a.x, f1 = a.z >= a.w || a.z != a.z;
    const bool f2 = b.x >= b.y || b.x != b.x, f3 = b.z >= b.w || b.z != b.z;
    *reinterpret_cast<float4*>(dx + 8 * v) = make_float4(f0 ? g.x : 0.f, f0 ? 0.f : g.x, f1 ? g.y : 0.f, f1 ? 0.f : g.y);
    *reinterpret_cast<float4*>(dx + 8 * v + 4) = make_float4(f2 ? g.z : 0.f, f2 ? 0.f : g.z, f3 ? g.w : 0.f, f3 ? 0.f : g.w);
  }
}

// gradient goes to the arg-max (first element on ties, like torch); other positions 0
__global__ void __launch_bounds__(256) m2d_maxpool2_bwd_kernel(const float* x, const float* dy, float* dx,
                                                               size_t rows, int L, int Lout) {
  const size_t total = rows * L;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / L;
    const int p = (int)(i - r * L);
    const int j = p >> 1;
    float g = 0.f;
    if (j < Lout) {
      const float a = x[r * L + 2 * j], b = x[r * L + 2 * j + 1];
      const bool first = a >= b || a != a;  // NaN propagates through the first slot
      if (((p & 1) == 0) == first) g = dy[r * Lout + j];
    }
    dx[i] = g;
  }
}

// Upsample(scale_factor=2, mode="linear", align_corners=False):
//   src = max((j + 0.5) / 2 - 0.5, 0); i0 = floor(src); i1 = min(i0 + 1, L - 1); w = src - i0
// (C, ypitch: the output row r = (b, c) starts at b * ypitch + c * 2L - the result may be a channel block of a wider
// buffer; C == 0: dense rows)
__device__ __forceinline__ size_t m2d_up_row(size_t r, int L, int C, long long ypitch) {
  if (C <= 0) return r * 2 * (size_t)L;
  const size_t b = r / (size_t)C;
  return b * (size_t)ypitch + (r - b * (size_t)C) * 2 * (size_t)L;
}

__global__ void __launch_bounds__(256) m2d_upsample2_fwd_kernel(const float* x, float* y, size_t rows, int L, int C,
                                                                long long ypitch) {
  const int Lo = 2 * L;
  const size_t total = rows * Lo;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / Lo;
    const int j = (int)(i - r * Lo);
    float src = ((float)j + 0.5f) * 0.5f - 0.5f;
    if (src < 0.f) src = 0.f;
    const int i0 = (int)src;
    const int i1 = i0 + 1 < L ? i0 + 1 : L - 1;
    const float w1 = src - (float)i0;
    const float w0 = 1.0f - w1;
    y[m2d_up_row(r, L, C, ypitch) + j] = w0 * x[r * L + i0] + w1 * x[r * L + i1];
  }
}

// L % 4 == 0: a thread owns four input positions of a row (one 16-byte load + the two neighbours) and writes their
// eight outputs (two 16-byte stores), walking the rows with a fixed position - no division per element. Same
// expression per output as the kernel above (w0 * x[i0] + w1 * x[i1] with w in {1, 0.75, 0.25, 0}): same bits.
__global__ void __launch_bounds__(256) m2d_upsample2_fwd_vec_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                    size_t rows, int L, int C, long long ypitch) {
  const int rvl = L >> 2;  // vectors per row (<= 256: launcher-checked)
  const int rpb = 256 / rvl;
  const int pv = threadIdx.x % rvl, rib = threadIdx.x / rvl;
  if (rib >= rpb) return;
  const int p0 = 4 * pv;
#pragma unroll 4
  for (size_t r = (size_t)blockIdx.x * rpb + rib; r < rows; r += (size_t)gridDim.x * rpb) {
    const float* xr = x + r * L;
    const float4 c = *reinterpret_cast<const float4*>(xr + p0);
    const float lft = p0 > 0 ? xr[p0 - 1] : 0.f;
    const float rgt = p0 + 4 < L ? xr[p0 + 4] : c.w;  // i1 = min(i0 + 1, L - 1)
    float o[8];
    // even outputs j = 2p: p = 0 -> src clamps to 0: 1 * x[0] + 0 * x[min(1, L-1)]; p >= 1 -> 0.25 * x[p-1] + 0.75 * x[p]
    o[0] = p0 > 0 ? 0.25f * lft + 0.75f * c.x : 1.0f * c.x + 0.0f * (L > 1 ? c.y : c.x);
    o[2] = 0.25f * c.x + 0.75f * c.y;
    o[4] = 0.25f * c.y + 0.75f * c.z;
    o[6] = 0.25f * c.z + 0.75f * c.w;
    // odd outputs j = 2p + 1: 0.75 * x[p] + 0.25 * x[min(p + 1, L - 1)]
    o[1] = 0.75f * c.x + 0.25f * c.y;
    o[3] = 0.75f * c.y + 0.25f * c.z;
    o[5] = 0.75f * c.z + 0.25f * c.w;
    o[7] = 0.75f * c.w + 0.25f * rgt;
    float* yr = y + m2d_up_row(r, L, C, ypitch) + 2 * p0;
    *reinterpret_cast<float4*>(yr) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(yr + 4) = make_float4(o[4], o[5], o[6], o[7]);
  }
}

// any L <= 256: a thread owns ONE input position of a row (three scalar loads from the same lines) and writes its two
// outputs as one 8-byte store, walking the rows with a fixed position. Same expressions: same bits.
__global__ void __launch_bounds__(256) m2d_upsample2_fwd_row_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                    size_t rows, int L, int C, long long ypitch) {
  const int rpb = 256 / L;
  const int p = threadIdx.x % L, rib = threadIdx.x / L;
  if (rib >= rpb) return;
#pragma unroll 4
  for (size_t r = (size_t)blockIdx.x * rpb + rib; r < rows; r += (size_t)gridDim.x * rpb) {
    const float* xr = x + r * L;
    const float c = xr[p];
    const float lft = p > 0 ? xr[p - 1] : 0.f;
    const float rgt = p + 1 < L ? xr[p + 1] : c;
    float2 o;
    o.x = p > 0 ? 0.25f * lft + 0.75f * c : 1.0f * c + 0.0f * rgt;
    o.y = 0.75f * c + 0.25f * rgt;
    *reinterpret_cast<float2*>(y + m2d_up_row(r, L, C, ypitch) + 2 * p) = o;
  }
}

// Any L, C * L even: the pass in the FLAT index space of a sample. Output element 2 g, 2 g + 1 of a sample's flattened
// (C, 2L) block come from input element g of its flattened (C, L) block (and a neighbour in the same row), so a thread
// owns two consecutive inputs g, g + 1 (g even: one 8-byte load + the two neighbours) and writes their four outputs as
// ONE 16-byte store at flat offset 2 g - every lane of a wave stores 16 contiguous bytes next to its neighbour's, whatever
// L is (round 6: the U-Net's L = 25 and 50 went through 8-byte stores of the per-row kernel at 2.9 TB/s, L = 100 through
// two interleaved 16-byte stores per thread at 3.4; a plain copy of that size runs 5.3, tools/hbm_sizes.py). Rows enter
// only through the position p = g mod L (first / last element of a row). Same expression per output as the kernels
// above: same bits.
__global__ void __launch_bounds__(256) m2d_upsample2_fwd_flat_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                     unsigned pairs_per_sample, size_t pairs, int L,
                                                                     long long ypitch) {
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < pairs; v += (size_t)gridDim.x * 256) {
    const size_t b = v / pairs_per_sample;
    const unsigned g = 2u * (unsigned)(v - b * pairs_per_sample);   // flat input index inside the sample (even)
    const float* xs = x + b * (size_t)(2u * pairs_per_sample);
    const float2 c = *reinterpret_cast<const float2*>(xs + g);
    const unsigned p0 = g % (unsigned)L;
    const unsigned p1 = p0 + 1 == (unsigned)L ? 0u : p0 + 1;
    // neighbours inside the rows (clamped reads stay inside the sample: the selects below never use a clamped value)
    const float lft0 = p0 > 0 ? xs[g - 1] : 0.f;
    const float rgt1 = p1 + 1 < (unsigned)L ? xs[g + 2] : c.y;
    const float lft1 = c.x, rgt0 = p0 + 1 < (unsigned)L ? c.y : c.x;
    float4 o;
    o.x = p0 > 0 ? 0.25f * lft0 + 0.75f * c.x : 1.0f * c.x + 0.0f * rgt0;
    o.y = 0.75f * c.x + 0.25f * rgt0;
    // (p1 == 0: g + 1 starts the next row - its left neighbour is not c.x, and its right one is xs[g + 2] or itself)
    o.z = p1 > 0 ? 0.25f * lft1 + 0.75f * c.y : 1.0f * c.y + 0.0f * rgt1;
    o.w = 0.75f * c.y + 0.25f * rgt1;
    float* ys = y + (ypitch > 0 ? b * (size_t)ypitch : b * (size_t)(4u * pairs_per_sample));
    *reinterpret_cast<float4*>(ys + 2 * (size_t)g) = o;
  }
}

// transpose of the interpolation: dx[i] = sum_j coef(j -> i) dy[j]; each input position
// receives from at most 4 outputs (2i-1 .. 2i+2), gathered here so there are no atomics.
__global__ void __launch_bounds__(256) m2d_upsample2_bwd_kernel(const float* dy, float* dx, size_t rows, int L) {
  const int Lo = 2 * L;
  const size_t total = rows * L;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / L;
    const int p = (int)(i - r * L);
    float g = 0.f;
    for (int j = 2 * p - 2; j <= 2 * p + 2; ++j) {
      if (j < 0 || j >= Lo) continue;
      float src = ((float)j + 0.5f) * 0.5f - 0.5f;
      if (src < 0.f) src = 0.f;
      const int i0 = (int)src;
      const int i1 = i0 + 1 < L ? i0 + 1 : L - 1;
      const float w1 = src - (float)i0;
      const float d = dy[r * Lo + j];
      if (i0 == p) g += (1.0f - w1) * d;
      if (i1 == p) g += w1 * d;
    }
    dx[i] = g;
  }
}

// (rows * L) % 4 == 0: a thread owns four consecutive elements q .. q + 3 of the flattened dx (16-byte store); their own
// output pairs are dy[2 q .. 2 q + 7] of the flattened dy (two 16-byte loads) plus one neighbour on either side. The
// terms of an element are added in the order of the loop above (j = 2p - 2 .. 2p + 2): same bits.
__global__ void __launch_bounds__(256) m2d_upsample2_bwd_flat_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                     size_t quads, int L) {
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < quads; v += (size_t)gridDim.x * 256) {
    const size_t q = 4 * v;
    const float4 a = *reinterpret_cast<const float4*>(dy + 2 * q);
    const float4 b = *reinterpret_cast<const float4*>(dy + 2 * q + 4);
    const float d[10] = {q > 0 ? dy[2 * q - 1] : 0.f, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w,
                         v + 1 < quads ? dy[2 * q + 8] : 0.f};   // d[1 + k] = dy[2 q + k]
    unsigned p = (unsigned)(q % (size_t)L);
    const float dm2 = p == 1 ? dy[2 * q - 2] : 0.f;       // (element 0's j = 0 when it is the second of its row)
    float4 o;
    float* op = &o.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ev = d[1 + 2 * e], od = d[2 + 2 * e];   // dy[2p], dy[2p + 1] of this element's row
      float g = 0.f;
      if (p == 1) g += 0.0f * (e == 0 ? dm2 : d[e == 0 ? 0 : 2 * e - 1]);   // j = 0 of the row reaches i1 = 1 with weight 0
      if (p >= 1) g += 0.25f * d[2 * e];                  // j = 2p - 1
      g += p >= 1 ? 0.75f * ev : 1.0f * ev;               // j = 2p
      if (p == 0 && L == 1) g += 0.0f * ev;
      g += 0.75f * od;                                     // j = 2p + 1
      if (p + 1 == (unsigned)L) g += 0.25f * od;           //   (i1 clamps onto p)
      else g += 0.25f * d[3 + 2 * e];                      // j = 2p + 2
      op[e] = g;
      p = p + 1 == (unsigned)L ? 0u : p + 1;
    }
    *reinterpret_cast<float4*>(dx + q) = o;
  }
}

static unsigned grid_for(size_t n, unsigned cap = 2048) {
  size_t b = (n + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

extern "C" {

// losses.py:13-20 — interpolation between real and fake with one alpha per sample.
int m2d_gp_interpolate(const float* real, const float* fake, const float* alpha, float* out, int B, int n,
                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || n <= 0) M2D_FAIL(M2D_ERR_ARG, "m2d_gp_interpolate: bad shape");
  M2dProfScope prof(M2D_FAM_POINTWISE, stream, 0.0, 12.0 * B * (double)n, "gp_interpolate");
  unsigned gx = grid_for((size_t)n, 64);
  hipLaunchKernelGGL(m2d_gp_interpolate_kernel, dim3(gx, B), dim3(256), 0, stream, real, fake, alpha, out, B, n);
  M2D_CHECK_LAUNCH("m2d_gp_interpolate_kernel");
  return M2D_OK;
}

// losses.py:47-60 — per-sample L2 norm of the critic's input gradient and the penalty.
// norms: B floats (saved for the backward), penalty: 1 float.
size_t m2d_gp_penalty_workspace_bytes(int B) { return (size_t)B * GP_SPLIT * sizeof(double); }

int m2d_gp_penalty_fwd(const float* g, float* norms, float* penalty, int B, int n, int lp, void* ws,
                       size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || n <= 0) M2D_FAIL(M2D_ERR_ARG, "m2d_gp_penalty_fwd: bad shape");
  if (!ws || ws_bytes < m2d_gp_penalty_workspace_bytes(B)) M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_gp_penalty_fwd: workspace");
  M2dProfScope prof(M2D_FAM_REDUCE, stream, 0.0, 4.0 * B * (double)n, "gp_norm_penalty", B, n, 0);
  hipLaunchKernelGGL(m2d_gp_norm_partial_kernel, dim3(GP_SPLIT, B), dim3(256), 0, stream, g, (double*)ws, n);
  hipLaunchKernelGGL(m2d_gp_penalty_kernel, dim3(1), dim3(256), 0, stream, (const double*)ws, norms, B, lp,
                     lp ? 0.0f : 1e-12f, penalty);
  M2D_CHECK_LAUNCH("m2d_gp_penalty_kernel");
  return M2D_OK;
}

int m2d_gp_penalty_bwd(const float* g, const float* norms, const float* gout, float* dg, int B, int n, int lp,
                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || n <= 0) M2D_FAIL(M2D_ERR_ARG, "m2d_gp_penalty_bwd: bad shape");
  M2dProfScope prof(M2D_FAM_POINTWISE, stream, 0.0, 8.0 * B * (double)n, "gp_penalty_bwd");
  unsigned gx = grid_for((size_t)n, 64);
  hipLaunchKernelGGL(m2d_gp_penalty_bwd_kernel, dim3(gx, B), dim3(256), 0, stream, g, norms, gout, dg, B, n, lp);
  M2D_CHECK_LAUNCH("m2d_gp_penalty_bwd_kernel");
  return M2D_OK;
}

size_t m2d_reduce_workspace_bytes(void) { return 2048 * sizeof(double); }

// torch.nn.L1Loss(reduction='mean') (phase3/train.py:170,226): out = mean |a - b|
int m2d_l1_mean_fwd(const float* a, const float* b, float* out, size_t n, void* ws, size_t ws_bytes,
                    void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0) M2D_FAIL(M2D_ERR_ARG, "m2d_l1_mean_fwd: empty");
  if (!ws || ws_bytes < m2d_reduce_workspace_bytes()) M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_l1_mean_fwd: workspace");
  M2dProfScope prof(M2D_FAM_REDUCE, stream, 0.0, 8.0 * (double)n, "l1_mean_fwd");
  const unsigned gx = grid_for(n);
  hipLaunchKernelGGL(m2d_absdiff_partial_kernel, dim3(gx), dim3(256), 0, stream, a, b, n, (double*)ws);
  hipLaunchKernelGGL(m2d_finish_sum_kernel, dim3(1), dim3(256), 0, stream, (const double*)ws, (int)gx,
                     1.0 / (double)n, out);
  M2D_CHECK_LAUNCH("m2d_l1_mean_fwd");
  return M2D_OK;
}

int m2d_l1_mean_bwd(const float* a, const float* b, const float* gout, float* da, size_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  M2dProfScope prof(M2D_FAM_POINTWISE, stream, 0.0, 12.0 * (double)n, "l1_mean_bwd");
  hipLaunchKernelGGL(m2d_absdiff_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, a, b, gout, da, n);
  M2D_CHECK_LAUNCH("m2d_l1_mean_bwd");
  return M2D_OK;
}

// losses.py:76-82 tv_loss on a (B, C, T) tensor given by element strides.
int m2d_tv_mean_fwd(const float* x, float* out, int B, int C, int T, long sb, long sc, long st, void* ws,
                    size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || C <= 0 || T < 2) M2D_FAIL(M2D_ERR_ARG, "m2d_tv_mean_fwd: bad shape");
  if (!ws || ws_bytes < m2d_reduce_workspace_bytes()) M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_tv_mean_fwd: workspace");
  const size_t total = (size_t)B * C * (T - 1);
  M2dProfScope prof(M2D_FAM_REDUCE, stream, 0.0, 4.0 * (double)total, "tv_mean_fwd");
  const unsigned gx = grid_for(total);
  hipLaunchKernelGGL(m2d_tv_partial_kernel, dim3(gx), dim3(256), 0, stream, x, B, C, T, sb, sc, st, (double*)ws);
  hipLaunchKernelGGL(m2d_finish_sum_kernel, dim3(1), dim3(256), 0, stream, (const double*)ws, (int)gx,
                     1.0 / (double)total, out);
  M2D_CHECK_LAUNCH("m2d_tv_mean_fwd");
  return M2D_OK;
}

int m2d_tv_mean_bwd(const float* x, const float* gout, float* dx, int B, int C, int T, long sb, long sc,
                    long st, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || C <= 0 || T < 2) M2D_FAIL(M2D_ERR_ARG, "m2d_tv_mean_bwd: bad shape");
  const size_t total = (size_t)B * C * T;
  M2dProfScope prof(M2D_FAM_POINTWISE, stream, 0.0, 8.0 * (double)total, "tv_mean_bwd");
  hipLaunchKernelGGL(m2d_tv_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, gout, dx, B, C, T, sb, sc, st);
  M2D_CHECK_LAUNCH("m2d_tv_mean_bwd");
  return M2D_OK;
}

// losses.py:85-89 jerkiness on a (B, C, T) tensor given by element strides (an evaluation metric of the
// reference, phase3/test.py:78-104): out = sum over channels of the squared third difference, mean over (B, T-3).
int m2d_jerk_mean_fwd(const float* x, float* out, int B, int C, int T, long sb, long sc, long st, void* ws,
                      size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || C <= 0 || T < 4) M2D_FAIL(M2D_ERR_ARG, "m2d_jerk_mean_fwd: bad shape (needs T >= 4)");
  if (!ws || ws_bytes < m2d_reduce_workspace_bytes()) M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_jerk_mean_fwd: workspace");
  const size_t total = (size_t)B * C * (T - 3);
  M2dProfScope prof(M2D_FAM_REDUCE, stream, 0.0, 4.0 * (double)B * C * T, "jerk_mean_fwd");
  const unsigned gx = grid_for(total);
  hipLaunchKernelGGL(m2d_jerk_partial_kernel, dim3(gx), dim3(256), 0, stream, x, B, C, T, sb, sc, st, (double*)ws);
  hipLaunchKernelGGL(m2d_finish_sum_kernel, dim3(1), dim3(256), 0, stream, (const double*)ws, (int)gx,
                     1.0 / ((double)B * (double)(T - 3)), out);
  M2D_CHECK_LAUNCH("m2d_jerk_mean_fwd");
  return M2D_OK;
}

// MinMaxScaler.transform (scale_, min_) / inverse_transform (1/scale_, -min_/scale_) over (rows, cols) features
// (utils.py:26-31,79-85; phase3/test.py:92-101): y = x * scale[col] + shift[col]. In place when y == x.
int m2d_affine_cols(const float* x, const float* scale, const float* shift, float* y, size_t rows, int cols,
                    void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (cols <= 0) M2D_FAIL(M2D_ERR_ARG, "m2d_affine_cols: bad shape");
  if (rows == 0) return M2D_OK;
  M2dProfScope prof(M2D_FAM_POINTWISE, stream, 0.0, 8.0 * (double)rows * cols, "affine_cols");
  hipLaunchKernelGGL(m2d_affine_cols_kernel, dim3(grid_for(rows * (size_t)cols)), dim3(256), 0, stream, x, scale,
                     shift, y, rows, cols);
  M2D_CHECK_LAUNCH("m2d_affine_cols");
  return M2D_OK;
}

// torch.optim.Adam's step over many tensors (see include/m2d.h); `items` is a HOST array
int m2d_adam_multi(const M2dAdamItem* items, int n, float lr, float beta1, float beta2, float eps, float bias_corr1,
                   float bias_corr2_sqrt, const float* skip, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || (n > 0 && !items) || !(bias_corr1 > 0.f) || !(bias_corr2_sqrt > 0.f))
    M2D_FAIL(M2D_ERR_ARG, "m2d_adam_multi: bad arguments");
  double bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    const M2dAdamItem& t = items[i];
    if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq || t.numel < 0)
      M2D_FAIL(M2D_ERR_ARG, "m2d_adam_multi: item %d has a NULL tensor", i);
    if ((t.pack_fwd || t.pack_bwd) && (t.cout <= 0 || t.cin <= 0 || t.ks <= 0 || (long long)t.cout * t.cin * t.ks != t.numel))
      M2D_FAIL(M2D_ERR_ARG, "m2d_adam_multi: item %d: packed images need the (cout, cin, ks) of the weight", i);
    bytes += 28.0 * (double)t.numel + 4.0 * (double)t.numel * ((t.pack_fwd ? 1 : 0) + (t.pack_bwd ? 1 : 0));
  }
  M2dProfScope prof(M2D_FAM_POINTWISE, stream, 0.0, bytes, "adam_multi");
  for (int i0 = 0; i0 < n;) {
    M2dAdamBatch b;
    b.n = 0;
    int blocks = 0;
    for (; i0 < n && b.n < M2D_ADAM_BATCH; ++i0) {
      if (items[i0].numel == 0) continue;
      M2dAdamItem& t = b.it[b.n];
      t = items[i0];
      const bool al = ((((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15u) == 0);
      const bool packed = t.pack_fwd || t.pack_bwd;
      const bool tile = packed && al && ((t.cin * t.ks) % 4) == 0;
      t.reserved = tile ? M2D_ADAM_TILE : ((al && !packed) ? M2D_ADAM_VEC : 0);
      b.first_block[b.n] = blocks;
      if (tile)
        blocks += ((t.cout + M2D_ADAM_TCO - 1) / M2D_ADAM_TCO) * ((t.cin * t.ks + M2D_ADAM_TJ - 1) / M2D_ADAM_TJ);
      else
        blocks += (int)((t.numel + M2D_ADAM_CHUNK - 1) / M2D_ADAM_CHUNK);
      ++b.n;
    }
    b.first_block[b.n] = blocks;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(m2d_adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, b, lr / bias_corr1, beta1, beta2,
                       eps, bias_corr2_sqrt, skip);
    M2D_CHECK_LAUNCH("m2d_adam_multi");
  }
  return M2D_OK;
}

// nn.Tanh of the 'tanh' heads and its derivatives (see the kernels): closed under differentiation like the conv ops
int m2d_tanh_fwd(const float* x, float* y, size_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0) return M2D_OK;
  hipLaunchKernelGGL(m2d_tanh_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, n);
  M2D_CHECK_LAUNCH("m2d_tanh_fwd");
  return M2D_OK;
}
int m2d_tanh_bwd(const float* gy, const float* y, float* gx, size_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0) return M2D_OK;
  hipLaunchKernelGGL(m2d_tanh_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, gy, y, gx, n);
  M2D_CHECK_LAUNCH("m2d_tanh_bwd");
  return M2D_OK;
}
int m2d_tanh_bwd_bwd(const float* g, const float* gy, const float* y, float* g_y, size_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0) return M2D_OK;
  hipLaunchKernelGGL(m2d_tanh_bwd_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, g, gy, y, g_y, n);
  M2D_CHECK_LAUNCH("m2d_tanh_bwd_bwd");
  return M2D_OK;
}

// [interpolated | real | fake] poses channels-first (3B, C, T) from real (B, T, C), fake rows (B*T, C), alpha (B,):
// replaces permute(0,2,1).contiguous() x2 (phase3/train.py:196-199), the interpolation (losses.py:13-25) and the
// concatenation of the batches the critic scores in one pass.
int m2d_pose_pack3(const float* real, const float* fake, const float* alpha, float* out, int B, int T, int C,
                   void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || T <= 0 || C <= 0 || C > 1024) M2D_FAIL(M2D_ERR_ARG, "m2d_pose_pack3: bad shape");
  M2dProfScope prof(M2D_FAM_POINTWISE, stream, 0.0, 20.0 * (double)B * T * C, "pose_pack3");
  const size_t lds = 2 * (size_t)M2D_PACK_TT * (C + 1) * sizeof(float);
  hipLaunchKernelGGL(m2d_pose_pack3_kernel, dim3(m2d_ceil_div(T, M2D_PACK_TT), B), dim3(256), lds, stream, real, fake,
                     alpha, out, B, T, C);
  M2D_CHECK_LAUNCH("m2d_pose_pack3");
  return M2D_OK;
}

// out[0..2] = (loss_critic, gp, w_dist) from scores (3B,) = [interpolated | real | fake] and the penalty term(s)
// pen0 (+ pen1, optional: the audio term, losses.py:56-60).
int m2d_wgan_critic_loss(const float* scores, int B, const float* pen0, const float* pen1, float gamma, float* out,
                         void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || !pen0) M2D_FAIL(M2D_ERR_ARG, "m2d_wgan_critic_loss: bad arguments");
  hipLaunchKernelGGL(m2d_wgan_critic_loss_kernel, dim3(1), dim3(256), 0, stream, scores, B, pen0, pen1, gamma, out);
  M2D_CHECK_LAUNCH("m2d_wgan_critic_loss");
  return M2D_OK;
}

// nn.MaxPool1d(2, 2) on (rows = B*C, L) (phase3/archis/default.py:235)
int m2d_maxpool2_fwd(const float* x, float* y, size_t rows, int L, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int Lout = L / 2;
  if (rows == 0 || Lout <= 0) M2D_FAIL(M2D_ERR_ARG, "m2d_maxpool2_fwd: bad shape");
  M2dProfScope prof(M2D_FAM_POINTWISE, stream, 0.0, 6.0 * rows * (double)L, "maxpool2_fwd");
  if (L == 2 * Lout && (rows * Lout) % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15u) == 0)
    hipLaunchKernelGGL(m2d_maxpool2_fwd_flat_kernel, dim3(grid_for(rows * Lout / 4, 4096)), dim3(256), 0, stream, x, y,
                       rows * Lout / 4, (size_t)0, 0LL);
  else
    hipLaunchKernelGGL(m2d_maxpool2_fwd_kernel, dim3(grid_for(rows * Lout, 4096)), dim3(256), 0, stream, x, y, rows, L, Lout);
  M2D_CHECK_LAUNCH("m2d_maxpool2_fwd");
  return M2D_OK;
}
// x: (B, C, L) whose samples are x_batch_stride floats apart (a channel block of a wider buffer, see
// m2d_upsample2_fwd_to); y dense (B, C, L / 2). L even, C * L a multiple of 8, 16-byte aligned sample starts.
int m2d_maxpool2_fwd_from(const float* x, float* y, size_t B, int C, int L, long long x_batch_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B == 0 || C <= 0 || L <= 0) M2D_FAIL(M2D_ERR_ARG, "m2d_maxpool2_fwd: bad shape");
  const long long cl = (long long)C * L;
  if (x_batch_stride <= 0 || x_batch_stride == cl) return m2d_maxpool2_fwd(x, y, B * (size_t)C, L, stream_);
  if ((L & 1) || (cl & 7) || x_batch_stride < cl || (x_batch_stride & 3) || (((uintptr_t)x | (uintptr_t)y) & 15u))
    M2D_FAIL(M2D_ERR_ARG, "m2d_maxpool2_fwd_from: L even, C * L %% 8 == 0, 16-byte aligned samples");
  const size_t rows = B * (size_t)C;
  M2dProfScope prof(M2D_FAM_POINTWISE, stream, 0.0, 6.0 * rows * (double)L, "maxpool2_fwd");
  const size_t nvec = rows * (size_t)(L / 2) / 4;
  hipLaunchKernelGGL(m2d_maxpool2_fwd_flat_kernel, dim3(grid_for(nvec, 4096)), dim3(256), 0, stream, x, y, nvec,
                     (size_t)(cl / 8), x_batch_stride);
  M2D_CHECK_LAUNCH("m2d_maxpool2_fwd");
  return M2D_OK;
}

int m2d_maxpool2_bwd(const float* x, const float* dy, float* dx, size_t rows, int L, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int Lout = L / 2;
  if (rows == 0 || Lout <= 0) M2D_FAIL(M2D_ERR_ARG, "m2d_maxpool2_bwd: bad shape");
  M2dProfScope prof(M2D_FAM_POINTWISE, stream, 0.0, 10.0 * rows * (double)L, "maxpool2_bwd");
  if (L == 2 * Lout && (rows * Lout) % 4 == 0 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15u) == 0)
    hipLaunchKernelGGL(m2d_maxpool2_bwd_flat_kernel, dim3(grid_for(rows * Lout / 4, 4096)), dim3(256), 0, stream, x, dy, dx,
                       rows * Lout / 4);
  else
    hipLaunchKernelGGL(m2d_maxpool2_bwd_kernel, dim3(grid_for(rows * L, 4096)), dim3(256), 0, stream, x, dy, dx, rows, L, Lout);
  M2D_CHECK_LAUNCH("m2d_maxpool2_bwd");
  return M2D_OK;
}

// nn.Upsample(scale_factor=2, mode="linear", align_corners=False) (phase3/archis/default.py:236)
// x (B, C, L) -> y: sample b's channels at y + b * y_batch_stride (y_batch_stride 0 or C * 2L: dense (B, C, 2L)); a larger
// stride writes into a channel block of a wider buffer (the U-Net's skip concatenation made in place)
int m2d_upsample2_fwd_to(const float* x, float* y, size_t B, int C, int L, long long y_batch_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B == 0 || C <= 0 || L <= 0) M2D_FAIL(M2D_ERR_ARG, "m2d_upsample2_fwd: bad shape");
  const size_t rows = B * (size_t)C;
  const bool dense = y_batch_stride <= 0 || y_batch_stride == (long long)C * 2 * L;
  if (!dense && y_batch_stride < (long long)C * 2 * L) M2D_FAIL(M2D_ERR_ARG, "m2d_upsample2_fwd: output batch stride smaller than a sample");
  const int Cp = dense ? 0 : C;
  const long long yp = dense ? 0 : y_batch_stride;
  M2dProfScope prof(M2D_FAM_POINTWISE, stream, 0.0, 12.0 * rows * (double)L, "upsample2_fwd");
  static const bool flat_on = [] { const char* e = getenv("M2D_UPSAMPLE_FLAT"); return !(e && e[0] == '0'); }();   // A/B lever
  const size_t per_sample = (size_t)C * L;
  if (flat_on && (per_sample & 1) == 0 && per_sample < (1u << 30) && (((uintptr_t)x | (uintptr_t)y) & 15u) == 0 && (yp & 3) == 0 &&
      true) {
    // (16-byte stores at flat offset 2 g, g even: aligned when every sample's block starts 16-byte aligned - dense blocks of
    // 2 C L floats with C L even, or a pitch that is a multiple of 4)
    const size_t pairs = B * (per_sample / 2);
    hipLaunchKernelGGL(m2d_upsample2_fwd_flat_kernel, dim3(grid_for(pairs, 16384)), dim3(256), 0, stream, x, y,
                       (unsigned)(per_sample / 2), pairs, L, yp);
  } else if ((L & 3) == 0 && L <= 1024 && (((uintptr_t)x | (uintptr_t)y) & 15u) == 0 && (yp & 3) == 0) {
    const size_t rpb = 256 / (L >> 2);
    hipLaunchKernelGGL(m2d_upsample2_fwd_vec_kernel, dim3(grid_for((rows + rpb - 1) / rpb * 256, 4096)), dim3(256), 0, stream,
                       x, y, rows, L, Cp, yp);
  } else if (L <= 256 && ((uintptr_t)y & 7u) == 0 && (yp & 1) == 0) {
    const size_t rpb = 256 / L;
    hipLaunchKernelGGL(m2d_upsample2_fwd_row_kernel, dim3(grid_for((rows + rpb - 1) / rpb * 256, 8192)), dim3(256), 0, stream,
                       x, y, rows, L, Cp, yp);
  } else {
    hipLaunchKernelGGL(m2d_upsample2_fwd_kernel, dim3(grid_for(rows * 2 * L, 4096)), dim3(256), 0, stream, x, y, rows, L, Cp, yp);
  }
  M2D_CHECK_LAUNCH("m2d_upsample2_fwd");
  return M2D_OK;
}
int m2d_upsample2_fwd(const float* x, float* y, size_t rows, int L, void* stream_) {
  return m2d_upsample2_fwd_to(x, y, rows, 1, L, 0, stream_);
}

int m2d_upsample2_bwd(const float* dy, float* dx, size_t rows, int L, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (rows == 0 || L <= 0) M2D_FAIL(M2D_ERR_ARG, "m2d_upsample2_bwd: bad shape");
  M2dProfScope prof(M2D_FAM_POINTWISE, stream, 0.0, 12.0 * rows * (double)L, "upsample2_bwd");
  static const bool flat_on = [] { const char* e = getenv("M2D_UPSAMPLE_FLAT"); return !(e && e[0] == '0'); }();   // A/B lever
  if (flat_on && ((rows * (size_t)L) & 3) == 0 && (((uintptr_t)dy | (uintptr_t)dx) & 15u) == 0)
    hipLaunchKernelGGL(m2d_upsample2_bwd_flat_kernel, dim3(grid_for(rows * L / 4, 16384)), dim3(256), 0, stream, dy, dx,
                       rows * L / 4, L);
  else
    hipLaunchKernelGGL(m2d_upsample2_bwd_kernel, dim3(grid_for(rows * L, 4096)), dim3(256), 0, stream, dy, dx, rows, L);
  M2D_CHECK_LAUNCH("m2d_upsample2_bwd");
  return M2D_OK;
}

}  // extern "C"
