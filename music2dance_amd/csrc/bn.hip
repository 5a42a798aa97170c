// BatchNorm1d (training and eval forward, training backward) on (B, C, L) / (B, C) fp32
// tensors, plus the per-channel sum used for conv / linear bias gradients.
//
// Reference semantics: nn.BatchNorm1d(eps=1e-5, momentum=0.1) exactly as
// phase3/archis/default.py:65,68,118-127,154,179-180,217 uses it:
//   train: y = gamma * (x - mean_b) / sqrt(var_biased + eps) + beta,
//          running_mean <- (1-m) rm + m mean_b, running_var <- (1-m) rv + m var_unbiased
//   eval : the same with the running statistics.
// These kernels are HBM-bound: the statistics pass reads x once (coalesced along the
// contiguous (c,l) run of every sample row, whatever L is), the apply pass reads x once
// and writes y once with the activation (ReLU / LeakyReLU) and an optional residual add
// fused. Partial sums are combined in fp64 so the result does not depend on the grid.
#include "m2d_common.h"

__device__ __forceinline__ void bn_divmod(int n, int d, float inv, int& q, int& r) {
  q = (int)((float)n * inv);
  r = n - q * d;
  if (r < 0) {
    q -= 1;
    r += d;
  } else if (r >= d) {
    q += 1;
    r -= d;
  }
}

// mode 0: (x, x*x)                      forward statistics
// mode 1: (dz, dz * xhat)               backward reductions, dz = dy * act'(gamma*xhat+beta)
// mode 2: (x * (mask>0 ? 1 : slope), 0) masked channel sum (bias gradients)
struct BnReduceArgs {
  const float* x;
  const float* dy;     // mode 1
  const float* mask;   // mode 2 (optional)
  const float* gamma;  // mode 1
  const float* beta;   // mode 1
  const float* mean;   // mode 1
  const float* invstd; // mode 1
  double* acc;         // [C][2]
  int B, C, L;
  int cpb;             // channels per block
  int rows_per_block;
  int vec_rows;        // short rows read as 16-byte pieces, four samples per step (launch_reduce decides)
  int mode, act;
  float slope;
  // Self-cleaning accumulation (callers that pass a zero-kept scratch, m2d_bn_scratch_bytes): `acc` is that scratch,
  // `ticket` its arrival counter; the block that arrives last reads the totals, finishes the op (fin) and leaves
  // scratch and ticket zeroed for the next launch on the stream - no memset and no finalize launch.
  unsigned* ticket;
  int fin;             // 1: totals -> sums_out; 2: forward statistics; 3: backward means + parameter gradients;
                       // 4: float channel sums -> f_out
  double* sums_out;    // fin 1
  float* f0;           // fin 2: save_mean    fin 3: dgamma   fin 4: out
  float* f1;           // fin 2: save_invstd  fin 3: dbeta
  float* f2;           // fin 2: running_mean fin 3: s_dz
  float* f3;           // fin 2: running_var  fin 3: s_dzx
  double count;
  float eps, momentum;
};

// (the launches on ONE stream share a scratch; two streams need two)
__device__ __forceinline__ void bn_finish(const BnReduceArgs& a, int c, double s0, double s1) {
  if (a.fin == 1) {
    a.sums_out[2 * c] = s0;
    a.sums_out[2 * c + 1] = s1;
  } else if (a.fin == 2) {
    const double mu = s0 / a.count;
    double var = s1 / a.count - mu * mu;
    if (var < 0.0) var = 0.0;
    a.f0[c] = (float)mu;
    a.f1[c] = (float)(1.0 / sqrt(var + (double)a.eps));
    if (a.f2) {
      const double unbiased = a.count > 1.0 ? var * (a.count / (a.count - 1.0)) : var;
      a.f2[c] = (float)((1.0 - a.momentum) * (double)a.f2[c] + a.momentum * mu);
      a.f3[c] = (float)((1.0 - a.momentum) * (double)a.f3[c] + a.momentum * unbiased);
    }
  } else if (a.fin == 3) {
    a.f1[c] = (float)s0;
    a.f0[c] = (float)s1;
    a.f2[c] = (float)(s0 / a.count);
    a.f3[c] = (float)(s1 / a.count);
  } else if (a.fin == 4) {
    a.f0[c] = (float)s0;
  }
}

// one element's contribution to the two running sums
__device__ __forceinline__ void bn_accum(const BnReduceArgs& a, float xv, float dyv, float mv, float g, float bt,
                                         float mu, float is, float& s0, float& s1) {
  if (a.mode == 0) {
    s0 += xv;
    s1 += xv * xv;
  } else if (a.mode == 1) {
    const float xh = (xv - mu) * is;
    float dz = dyv;
    if (a.act) {
      const float z = g * xh + bt;
      if (!(z > 0.f)) dz = a.act == 1 ? 0.f : dz * a.slope;
    }
    s0 += dz;
    s1 += dz * xh;
  } else {
    s0 += a.mask ? xv * (mv > 0.f ? 1.f : a.slope) : xv;
  }
}

__global__ void __launch_bounds__(256) m2d_bn_reduce_kernel(const BnReduceArgs a) {
  __shared__ double sh0[256];
  __shared__ double sh1[256];
  const int t = threadIdx.x;
  const int c0 = blockIdx.x * a.cpb;
  const int nch = (a.C - c0) < a.cpb ? (a.C - c0) : a.cpb;
  const int P = nch * a.L;  // contiguous positions of this block inside one sample row
  const int n_begin = blockIdx.y * a.rows_per_block;
  int n_end = n_begin + a.rows_per_block;
  if (n_end > a.B) n_end = a.B;
  const size_t row_stride = (size_t)a.C * a.L;
  float s0 = 0.f, s1 = 0.f;
  if (a.cpb == 1 && (a.L & 3) == 0) {
    // long rows (the audio critic's activations: L up to 19 200): one channel per block, 16-byte
    // loads, four positions per thread and step
    float g = 0.f, bt = 0.f, mu = 0.f, is = 0.f;
    if (a.mode == 1) {
      g = a.gamma[c0];
      bt = a.beta[c0];
      mu = a.mean[c0];
      is = a.invstd[c0];
    }
    float q0[4] = {0.f, 0.f, 0.f, 0.f}, q1[4] = {0.f, 0.f, 0.f, 0.f};
    // the block's (rows x L/4) 16-byte pieces of this channel as ONE index space: with a loop over rows outside, a row
    // of L = 200 kept 50 of the 256 threads busy (the U-Net's (4 800, 128, 200) backward sums ran at 1.6 TB/s)
    const int L4 = P >> 2;
    const float L4_inv = 1.0f / (float)L4;
    const int total = (n_end - n_begin) * L4;
#pragma unroll 4
    for (int i = t; i < total; i += 256) {
      int nn, p4;
      bn_divmod(i, L4, L4_inv, nn, p4);
      const size_t rb = (size_t)(n_begin + nn) * row_stride + (size_t)c0 * a.L;
      const int p = 4 * p4;
      {
        const float4 xv = *reinterpret_cast<const float4*>(a.x + rb + p);
        float4 dv = make_float4(0.f, 0.f, 0.f, 0.f), mv = dv;
        if (a.mode == 1) dv = *reinterpret_cast<const float4*>(a.dy + rb + p);
        if (a.mode == 2 && a.mask) mv = *reinterpret_cast<const float4*>(a.mask + rb + p);
        bn_accum(a, xv.x, dv.x, mv.x, g, bt, mu, is, q0[0], q1[0]);
        bn_accum(a, xv.y, dv.y, mv.y, g, bt, mu, is, q0[1], q1[1]);
        bn_accum(a, xv.z, dv.z, mv.z, g, bt, mu, is, q0[2], q1[2]);
        bn_accum(a, xv.w, dv.w, mv.w, g, bt, mu, is, q0[3], q1[3]);
      }
    }
    s0 = (q0[0] + q0[1]) + (q0[2] + q0[3]);
    s1 = (q1[0] + q1[1]) + (q1[2] + q1[3]);
  } else if (a.cpb > 1 && (a.L & 3) == 0 && (a.vec_rows != 0)) {
    // short rows, several channels per block (the decoder / encoder BatchNorms of (B*T, C, L <= 128) activations): the
    // block's P = cpb * L <= 256 contiguous floats of a sample are at most 64 16-byte pieces, so the 256 threads take
    // FOUR samples per step (thread = (sample t / 64, piece t % 64)), four steps in flight - round 6: the scalar walk
    // below read 4 bytes per lane and load (2.3-2.5 TB/s on the 126 MB backward sums of C3, the U-Net's L = 100 / 50)
    const int piece = t & 63, rsub = t >> 6;
    const int P4 = P >> 2;
    if (piece < P4) {
      const int c = c0 + (4 * piece) / a.L;
      float g = 0.f, bt = 0.f, mu = 0.f, is = 0.f;
      if (a.mode == 1) {
        g = a.gamma[c];
        bt = a.beta[c];
        mu = a.mean[c];
        is = a.invstd[c];
      }
      const size_t base = (size_t)c0 * a.L + 4 * piece;
      float q0[4] = {0.f, 0.f, 0.f, 0.f}, q1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int n = n_begin + rsub; n < n_end; n += 4) {
        const size_t idx = (size_t)n * row_stride + base;
        const float4 xv = *reinterpret_cast<const float4*>(a.x + idx);
        float4 dv = make_float4(0.f, 0.f, 0.f, 0.f), mv = dv;
        if (a.mode == 1) dv = *reinterpret_cast<const float4*>(a.dy + idx);
        if (a.mode == 2 && a.mask) mv = *reinterpret_cast<const float4*>(a.mask + idx);
        bn_accum(a, xv.x, dv.x, mv.x, g, bt, mu, is, q0[0], q1[0]);
        bn_accum(a, xv.y, dv.y, mv.y, g, bt, mu, is, q0[1], q1[1]);
        bn_accum(a, xv.z, dv.z, mv.z, g, bt, mu, is, q0[2], q1[2]);
        bn_accum(a, xv.w, dv.w, mv.w, g, bt, mu, is, q0[3], q1[3]);
      }
      s0 = (q0[0] + q0[1]) + (q0[2] + q0[3]);
      s1 = (q1[0] + q1[1]) + (q1[2] + q1[3]);
    }
  } else {
    for (int p = t; p < P; p += 256) {
      const int c = c0 + (a.cpb > 1 ? p / a.L : 0);
      float g = 0.f, bt = 0.f, mu = 0.f, is = 0.f;
      if (a.mode == 1) {
        g = a.gamma[c];
        bt = a.beta[c];
        mu = a.mean[c];
        is = a.invstd[c];
      }
      const size_t base = (size_t)c0 * a.L + p;
      // eight independent row chains keep eight loads per tensor in flight (the block's rows are
      // rows_per_block strided reads per thread: the loop is latency-bound)
      float q0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, q1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int n = n_begin;
      for (; n + 8 <= n_end; n += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const size_t idx = (size_t)(n + j) * row_stride + base;
          bn_accum(a, a.x[idx], a.mode == 1 ? a.dy[idx] : 0.f, (a.mode == 2 && a.mask) ? a.mask[idx] : 0.f, g, bt, mu,
                   is, q0[j], q1[j]);
        }
      }
      for (; n < n_end; ++n) {
        const size_t idx = (size_t)n * row_stride + base;
        bn_accum(a, a.x[idx], a.mode == 1 ? a.dy[idx] : 0.f, (a.mode == 2 && a.mask) ? a.mask[idx] : 0.f, g, bt, mu, is,
                 q0[0], q1[0]);
      }
      q0[0] = ((q0[0] + q0[1]) + (q0[2] + q0[3])) + ((q0[4] + q0[5]) + (q0[6] + q0[7]));
      q1[0] = ((q1[0] + q1[1]) + (q1[2] + q1[3])) + ((q1[4] + q1[5]) + (q1[6] + q1[7]));
      s0 += q0[0];
      s1 += q1[0];
    }
  }
  sh0[t] = (double)s0;
  sh1[t] = (double)s1;
  __syncthreads();
  if (a.cpb > 1) {
    // thread j < nch sums the L partials of its channel (L <= 256 / cpb)
    if (t < nch) {
      double r0 = 0.0, r1 = 0.0;
      if ((a.L & 3) == 0 && (a.vec_rows != 0)) {   // partials at [sample slot][piece]: the channel's L / 4 pieces x 4 slots
        const int L4 = a.L >> 2;
        for (int rs = 0; rs < 4; ++rs)
          for (int l = 0; l < L4; ++l) {
            r0 += sh0[rs * 64 + t * L4 + l];
            r1 += sh1[rs * 64 + t * L4 + l];
          }
      } else
      for (int l = 0; l < a.L; ++l) {
        r0 += sh0[t * a.L + l];
        r1 += sh1[t * a.L + l];
      }
      atomicAdd(&a.acc[2 * (c0 + t)], r0);
      if (a.mode != 2) atomicAdd(&a.acc[2 * (c0 + t) + 1], r1);
    }
  } else {
    for (int s = 128; s > 0; s >>= 1) {
      if (t < s) {
        sh0[t] += sh0[t + s];
        sh1[t] += sh1[t + s];
      }
      __syncthreads();
    }
    if (t == 0) {
      atomicAdd(&a.acc[2 * c0], sh0[0]);
      if (a.mode != 2) atomicAdd(&a.acc[2 * c0 + 1], sh1[0]);
    }
  }
  if (!a.ticket) return;
  // last block to arrive finishes the op and re-zeroes the scratch
  __shared__ int s_last;
  // this block's fp64 atomics must have been performed before its ticket is drawn. They execute at device scope (past
  // the XCD's L2), so waiting for their completion is enough; a release fence here (`__threadfence()`) also writes the
  // L2 back - once per block, a thousand times per launch: 52 -> 140 us on the 126 MB backward sums.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) {
    const unsigned n = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (n + 1u == gridDim.x * gridDim.y) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  for (int c = t; c < a.C; c += 256) {  // (device-scope loads: they do not read this XCD's L2)
    const double s0 = __hip_atomic_load(&a.acc[2 * c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double s1 = a.mode != 2 ? __hip_atomic_load(&a.acc[2 * c + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    a.acc[2 * c] = 0.0;
    a.acc[2 * c + 1] = 0.0;
    bn_finish(a, c, s0, s1);
  }
  if (t == 0) *a.ticket = 0u;
}

static int launch_reduce(BnReduceArgs& a, hipStream_t stream) {
  a.cpb = a.L >= 256 ? 1 : 256 / a.L;
  if (a.cpb < 1) a.cpb = 1;
  if (a.cpb > a.C) a.cpb = a.C;
  const int groups = m2d_ceil_div(a.C, a.cpb);
  int nsplit = m2d_ceil_div(1024, groups);
  // every block ends in one fp64 atomic pair per channel: with few channel groups (the decoder's (B*T, 256)
  // activations: ONE group) a thousand row slices meant a thousand serialised atomics per address - 40 us for 8 MB
  // (measured at (7680, 256, 1): 51 / 30 / 26 / 32 / 54 us with at most 1024 / 384 / 192 / 96 / 48 slices)
  if (a.cpb > 1 && nsplit > 192) nsplit = 192;
  if (nsplit > a.B) nsplit = a.B;
  if (nsplit > 65535) nsplit = 65535;
  if (nsplit < 1) nsplit = 1;
  a.rows_per_block = m2d_ceil_div(a.B, nsplit);
  {
    static const bool on = [] { const char* e = getenv("M2D_BN_VEC_ROWS"); return !(e && e[0] == '0'); }();   // A/B lever
    const uintptr_t al = (uintptr_t)a.x | (uintptr_t)(a.mode == 1 ? a.dy : a.x) | (uintptr_t)((a.mode == 2 && a.mask) ? a.mask : a.x);
    a.vec_rows = (on && a.cpb > 1 && (a.L & 3) == 0 && (al & 15u) == 0) ? 1 : 0;
  }
  nsplit = m2d_ceil_div(a.B, a.rows_per_block);
  if (!a.ticket && hipMemsetAsync(a.acc, 0, sizeof(double) * 2 * a.C, stream) != hipSuccess)
    M2D_FAIL(M2D_ERR_HIP, "bn reduce: memset failed");
  hipLaunchKernelGGL(m2d_bn_reduce_kernel, dim3(groups, nsplit), dim3(256), 0, stream, a);
  M2D_CHECK_LAUNCH("m2d_bn_reduce_kernel");
  return M2D_OK;
}

// ---- finalize kernels (one thread per channel) ---------------------------------
__global__ void m2d_bn_finalize_fwd_kernel(const double* acc, float* mean, float* invstd,
                                           float* running_mean, float* running_var, int C, double count,
                                           float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mu = acc[2 * c] / count;
  double var = acc[2 * c + 1] / count - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
    running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
    running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
  }
}

__global__ void m2d_bn_eval_stats_kernel(const float* running_mean, const float* running_var, float* mean,
                                         float* invstd, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = running_mean[c];
  invstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// acc: this process's sums (parameter gradients); acc_g / count: the sums and element count the batch
// statistics were taken over (the same buffer, or the all-reduced one under synchronised BatchNorm)
__global__ void m2d_bn_finalize_bwd_kernel(const double* acc, const double* acc_g, float* dgamma, float* dbeta,
                                           float* s_dz, float* s_dzx, int C, double count) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] = (float)acc[2 * c];
  dgamma[c] = (float)acc[2 * c + 1];
  s_dz[c] = (float)(acc_g[2 * c] / count);
  s_dzx[c] = (float)(acc_g[2 * c + 1] / count);
}

__global__ void m2d_acc_to_float_kernel(const double* acc, float* out, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) out[c] = (float)acc[2 * c];
}

// ---- elementwise apply ------------------------------------------------------------
struct BnApplyArgs {
  const float* x;
  const float* dy;        // backward only
  const float* residual;  // forward only (optional)
  const float* gamma;
  const float* beta;
  const float* mean;
  const float* invstd;
  const float* s_dz;      // backward: sum(dz)/count
  const float* s_dzx;     // backward: sum(dz*xhat)/count
  float* out;
  int C, L;
  float L_inv;
  int row_len;            // C*L
  long long out_pitch;    // elements between consecutive samples of `out` (>= row_len; the inputs are dense)
  int backward, act;
  float slope;
  float* pool_out;        // forward, optional: MaxPool1d(2, 2) of the result as well, dense (B, C, L / 2) (L even)
  // forward, optional: the finalisation of the batch statistics inside this launch (round 6: one launch less per
  // BatchNorm). fin_sums != NULL: mean / invstd of a thread's channels are computed from the fp64 sums in its prologue
  // (the expressions of m2d_bn_finalize_fwd_kernel: same bits), `mean` / `invstd` above are then OUTPUTS (save_mean /
  // save_invstd) written - with the running-buffer update - by workgroup (0, 0) alone; nobody reads them in this launch.
  const double* fin_sums;
  double fin_count;
  float fin_eps, fin_momentum;
  float* fin_running_mean;
  float* fin_running_var;
};

// mean / invstd of channel c: from the arrays, or from the batch sums (see BnApplyArgs::fin_sums)
__device__ __forceinline__ void bn_mean_invstd(const BnApplyArgs& a, int c, float& mu, float& is) {
  if (a.fin_sums) {
    const double m = a.fin_sums[2 * c] / a.fin_count;
    double var = a.fin_sums[2 * c + 1] / a.fin_count - m * m;
    if (var < 0.0) var = 0.0;
    mu = (float)m;
    is = (float)(1.0 / sqrt(var + (double)a.fin_eps));
  } else {
    mu = a.mean[c];
    is = a.invstd[c];
  }
}
// workgroup (0, 0): save_mean / save_invstd and the running buffers (what m2d_bn_finalize_fwd_kernel leaves behind)
__device__ __forceinline__ void bn_finalize_side(const BnApplyArgs& a) {
  if (!a.fin_sums || blockIdx.x != 0 || blockIdx.y != 0) return;
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    const double m = a.fin_sums[2 * c] / a.fin_count;
    double var = a.fin_sums[2 * c + 1] / a.fin_count - m * m;
    if (var < 0.0) var = 0.0;
    const_cast<float*>(a.mean)[c] = (float)m;
    const_cast<float*>(a.invstd)[c] = (float)(1.0 / sqrt(var + (double)a.fin_eps));
    if (a.fin_running_mean) {
      const double unbiased = a.fin_count > 1.0 ? var * (a.fin_count / (a.fin_count - 1.0)) : var;
      a.fin_running_mean[c] = (float)((1.0 - a.fin_momentum) * (double)a.fin_running_mean[c] + a.fin_momentum * m);
      a.fin_running_var[c] = (float)((1.0 - a.fin_momentum) * (double)a.fin_running_var[c] + a.fin_momentum * unbiased);
    }
  }
}

// A thread owns ONE vector position of the (C, L) row and walks the batch: its channel(s) - one when L % 4 == 0, four
// consecutive ones when L == 1 - and their parameters are fixed for the whole launch (registers), so the loop body is
// load / FMA / store. (Rounds 1-2 walked a flat index: a 64-bit modulo and up to four divmods + sixteen parameter loads
// per 16-byte vector - 4.5 TB/s. Rows shorter than 256 vectors put several rows in a block.)
template <int VEC>
__global__ void __launch_bounds__(256) m2d_bn_apply_kernel(const BnApplyArgs a, int B) {
  const int rvl = a.row_len / VEC;  // vectors per row
  const int tid = threadIdx.x;
  bn_finalize_side(a);
  int pv, r0, rstep;
  if (rvl >= 256) {
    pv = blockIdx.x * 256 + tid;
    if (pv >= rvl) return;
    r0 = blockIdx.y;
    rstep = gridDim.y;
  } else {
    const int rpb = 256 / rvl;
    pv = tid % rvl;
    const int rib = tid / rvl;
    if (rib >= rpb) return;
    r0 = blockIdx.y * rpb + rib;
    rstep = gridDim.y * rpb;
  }
  const int e0 = pv * VEC;
  const bool percol = a.L == 1;  // the VEC elements are VEC consecutive channels (else one channel)
  float g[VEC], bt[VEC], mu[VEC], is[VEC], sdz[VEC], sdzx[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = percol ? e0 + j : e0 / a.L;
    g[j] = a.gamma[c]; bt[j] = a.beta[c];
    if (j == 0 || percol) bn_mean_invstd(a, c, mu[j], is[j]);
    else { mu[j] = mu[0]; is[j] = is[0]; }
    sdz[j] = a.backward ? a.s_dz[c] : 0.f;
    sdzx[j] = a.backward ? a.s_dzx[c] : 0.f;
  }
#pragma unroll 2
  for (int r = r0; r < B; r += rstep) {
    const size_t idx = (size_t)r * a.row_len + e0;
    float xv[VEC], dv[VEC], rv[VEC], ov[VEC];
    if constexpr (VEC == 4) {
      const float4 t = *reinterpret_cast<const float4*>(a.x + idx);
      xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
      if (a.backward) {
        const float4 d = *reinterpret_cast<const float4*>(a.dy + idx);
        dv[0] = d.x; dv[1] = d.y; dv[2] = d.z; dv[3] = d.w;
      }
      if (a.residual) {
        const float4 q = *reinterpret_cast<const float4*>(a.residual + idx);
        rv[0] = q.x; rv[1] = q.y; rv[2] = q.z; rv[3] = q.w;
      }
    } else if constexpr (VEC == 2) {
      const float2 t = *reinterpret_cast<const float2*>(a.x + idx);
      xv[0] = t.x; xv[1] = t.y;
      if (a.backward) {
        const float2 d = *reinterpret_cast<const float2*>(a.dy + idx);
        dv[0] = d.x; dv[1] = d.y;
      }
      if (a.residual) {
        const float2 q = *reinterpret_cast<const float2*>(a.residual + idx);
        rv[0] = q.x; rv[1] = q.y;
      }
    } else {
      xv[0] = a.x[idx];
      if (a.backward) dv[0] = a.dy[idx];
      if (a.residual) rv[0] = a.residual[idx];
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float xh = (xv[j] - mu[j]) * is[j];
      const float z = g[j] * xh + bt[j];
      if (!a.backward) {
        float y = z;
        if (a.act == 1) y = z > 0.f ? z : 0.f;
        else if (a.act == 2) y = z > 0.f ? z : z * a.slope;
        if (a.residual) y += rv[j];
        ov[j] = y;
      } else {
        float dz = dv[j];
        if (a.act && !(z > 0.f)) dz = a.act == 1 ? 0.f : dz * a.slope;
        ov[j] = g[j] * is[j] * (dz - sdz[j] - xh * sdzx[j]);
      }
    }
    const size_t odx = (size_t)r * (size_t)a.out_pitch + e0;
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4*>(a.out + odx) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    } else if constexpr (VEC == 2) {
      *reinterpret_cast<float2*>(a.out + odx) = make_float2(ov[0], ov[1]);
    } else {
      a.out[odx] = ov[0];
    }
    if constexpr (VEC >= 2) {
      if (a.pool_out) {   // (a > b ? a : b, as m2d_maxpool2_fwd: the pairs of a row never straddle a vector - L is even)
        const size_t pdx = (size_t)r * (size_t)(a.row_len >> 1) + (e0 >> 1);
        if constexpr (VEC == 4)
          *reinterpret_cast<float2*>(a.pool_out + pdx) = make_float2(ov[0] > ov[1] ? ov[0] : ov[1], ov[2] > ov[3] ? ov[2] : ov[3]);
        else
          a.pool_out[pdx] = ov[0] > ov[1] ? ov[0] : ov[1];
      }
    }
  }
}

// BatchNorm apply + activation + Upsample(scale_factor=2, linear) in ONE pass (round 6; the U-Net's decoder levels without
// an autograd graph: the normalised tensor is consumed by the upsampling alone - writing and re-reading it was 2 of the 5
// tensor-sized transfers of the pair of passes). Flat index space of a sample as m2d_upsample2_fwd_flat_kernel
// (csrc/pointwise.hip): a thread owns two consecutive inputs g, g + 1 and writes their four outputs as one 16-byte store;
// here each input is normalised first ((x - mean) * invstd * gamma + beta, activation - the expression of
// m2d_bn_apply_kernel, so the result equals bn_apply followed by upsample bit for bit). The channel of an element follows
// from its flat index; the parameters come from the cache.
__global__ void __launch_bounds__(256) m2d_bn_upsample2_flat_kernel(const BnApplyArgs a, unsigned pairs_per_sample, size_t pairs) {
  const unsigned L = (unsigned)a.L;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < pairs; v += (size_t)gridDim.x * 256) {
    const size_t b = v / pairs_per_sample;
    const unsigned g = 2u * (unsigned)(v - b * pairs_per_sample);
    const float* xs = a.x + b * (size_t)a.row_len;
    const unsigned c0 = g / L, p0 = g - c0 * L;
    const bool wrap = p0 + 1 == L;                 // g + 1 starts the next row
    const unsigned p1 = wrap ? 0u : p0 + 1, c1 = wrap ? c0 + 1 : c0;
    const float g0 = a.gamma[c0], b0 = a.beta[c0], m0 = a.mean[c0], i0 = a.invstd[c0];
    const float g1 = a.gamma[c1], b1 = a.beta[c1], m1 = a.mean[c1], i1 = a.invstd[c1];
    auto norm = [&](float x, float gg, float bb, float mm, float ii) {
      const float xh = (x - mm) * ii;
      const float z = gg * xh + bb;
      float y = z;
      if (a.act == 1) y = z > 0.f ? z : 0.f;
      else if (a.act == 2) y = z > 0.f ? z : z * a.slope;
      return y;
    };
    const float2 raw = *reinterpret_cast<const float2*>(xs + g);
    const float n0 = norm(raw.x, g0, b0, m0, i0), n1 = norm(raw.y, g1, b1, m1, i1);
    const float lft0 = p0 > 0 ? norm(xs[g - 1], g0, b0, m0, i0) : 0.f;
    const float rgt1 = p1 + 1 < L ? norm(xs[g + 2], g1, b1, m1, i1) : n1;
    const float lft1 = n0, rgt0 = p0 + 1 < L ? n1 : n0;
    float4 o;
    o.x = p0 > 0 ? 0.25f * lft0 + 0.75f * n0 : 1.0f * n0 + 0.0f * rgt0;
    o.y = 0.75f * n0 + 0.25f * rgt0;
    o.z = p1 > 0 ? 0.25f * lft1 + 0.75f * n1 : 1.0f * n1 + 0.0f * rgt1;
    o.w = 0.75f * n1 + 0.25f * rgt1;
    float* ys = a.out + b * (size_t)a.out_pitch;
    *reinterpret_cast<float4*>(ys + 2 * (size_t)g) = o;
  }
}

// The same for even L in m2d_bn_apply_kernel's organisation: a thread owns TWO consecutive positions of the (C, L) row -
// one channel, its parameters in registers for the whole launch, the pair never straddles a row - and walks the batch:
// 8-byte load + two neighbours, four outputs as one 16-byte store. No division and no parameter load in the loop
// (the flat form above: 3.6-3.9 TB/s at the U-Net's L = 50 / 100 / 200; it stays for odd L).
__global__ void __launch_bounds__(256) m2d_bn_upsample2_rows_kernel(const BnApplyArgs a, int B) {
  const int rvl = a.row_len / 2;  // pairs per row
  const int tid = threadIdx.x;
  bn_finalize_side(a);
  int pv, r0, rstep;
  if (rvl >= 256) {
    pv = blockIdx.x * 256 + tid;
    if (pv >= rvl) return;
    r0 = blockIdx.y;
    rstep = gridDim.y;
  } else {
    const int rpb = 256 / rvl;
    pv = tid % rvl;
    const int rib = tid / rvl;
    if (rib >= rpb) return;
    r0 = blockIdx.y * rpb + rib;
    rstep = gridDim.y * rpb;
  }
  const int e0 = 2 * pv;
  const int c = e0 / a.L, p0 = e0 - c * a.L;
  const float gg = a.gamma[c], bb = a.beta[c];
  float mm, ii;
  bn_mean_invstd(a, c, mm, ii);
  const bool has_l = p0 > 0, has_r = p0 + 2 < a.L;
  auto norm = [&](float x) {
    const float xh = (x - mm) * ii;
    const float z = gg * xh + bb;
    float y = z;
    if (a.act == 1) y = z > 0.f ? z : 0.f;
    else if (a.act == 2) y = z > 0.f ? z : z * a.slope;
    return y;
  };
#pragma unroll 2
  for (int r = r0; r < B; r += rstep) {
    const float* xp = a.x + (size_t)r * a.row_len + e0;
    const float2 raw = *reinterpret_cast<const float2*>(xp);
    const float lraw = has_l ? xp[-1] : 0.f, rraw = has_r ? xp[2] : 0.f;
    const float n0 = norm(raw.x), n1 = norm(raw.y);
    const float lft0 = norm(lraw);
    const float rgt1 = has_r ? norm(rraw) : n1;
    float4 o;
    o.x = has_l ? 0.25f * lft0 + 0.75f * n0 : 1.0f * n0 + 0.0f * n1;
    o.y = 0.75f * n0 + 0.25f * n1;
    o.z = 0.25f * n0 + 0.75f * n1;
    o.w = 0.75f * n1 + 0.25f * rgt1;
    *reinterpret_cast<float4*>(a.out + (size_t)r * (size_t)a.out_pitch + 2 * (size_t)e0) = o;
  }
}

static int launch_apply(BnApplyArgs& a, int B, hipStream_t stream) {
  if (a.out_pitch <= 0) a.out_pitch = a.row_len;
  if (a.out_pitch < a.row_len) M2D_FAIL(M2D_ERR_ARG, "BatchNorm apply: output batch stride smaller than a sample");
  // (vector stores: the output's sample starts must keep the vector alignment too)
  const bool vec4 = (a.row_len % 4 == 0) && (a.L == 1 || a.L % 4 == 0) && (a.out_pitch % 4 == 0) && (((uintptr_t)a.out & 15) == 0);
  const bool vec2 = !vec4 && (a.row_len % 2 == 0) && (a.L == 1 || a.L % 2 == 0) && (a.out_pitch % 2 == 0) &&
                    (((uintptr_t)a.out & 7) == 0);  // e.g. the WaveGAN encoder's L = 794
  if (a.pool_out && ((a.L & 1) || !(vec4 || vec2) || a.backward))
    M2D_FAIL(M2D_ERR_ARG, "BatchNorm apply + max-pool: even L and 8-byte aligned rows");
  const int rvl = vec4 ? a.row_len / 4 : vec2 ? a.row_len / 2 : a.row_len;
  unsigned gx, gy;
  if (rvl >= 256) {
    gx = (unsigned)m2d_ceil_div(rvl, 256);
    gy = 4096u / gx;
    if (gy < 1) gy = 1;
    if (gy > (unsigned)B) gy = (unsigned)B;
  } else {
    const int rpb = 256 / rvl;
    gx = 1;
    gy = (unsigned)m2d_ceil_div(B, rpb);
    if (gy > 4096u) gy = 4096u;
  }
  if (gy > 65535u) gy = 65535u;
  if (vec4) hipLaunchKernelGGL(m2d_bn_apply_kernel<4>, dim3(gx, gy), dim3(256), 0, stream, a, B);
  else if (vec2) hipLaunchKernelGGL(m2d_bn_apply_kernel<2>, dim3(gx, gy), dim3(256), 0, stream, a, B);
  else hipLaunchKernelGGL(m2d_bn_apply_kernel<1>, dim3(gx, gy), dim3(256), 0, stream, a, B);
  M2D_CHECK_LAUNCH("m2d_bn_apply_kernel");
  return M2D_OK;
}

extern "C" {

// bytes of scratch every bn / channel-sum call needs (fp64 accumulators + 2 float rows)
size_t m2d_bn_workspace_bytes(int C) { return (size_t)C * (2 * sizeof(double) + 2 * sizeof(float)) + 64; }

// bytes of the optional zero-kept scratch (`scratch` arguments below): fp64 accumulators + the arrival counter. The
// caller zeroes it ONCE; every call that takes it leaves it zeroed. One scratch per stream (launches of one stream
// reuse it in order); with scratch == NULL the calls memset their accumulators and finalise in a second launch.
size_t m2d_bn_scratch_bytes(int C) { return (size_t)C * 2 * sizeof(double) + 64; }
static inline void bn_use_scratch(BnReduceArgs& r, void* scratch, int C) {
  r.acc = (double*)scratch;
  r.ticket = (unsigned*)((double*)scratch + 2 * (size_t)C);
}

static int bn_check(const char* who, int B, int C, int L) {
  if (B <= 0 || C <= 0 || L <= 0) M2D_FAIL(M2D_ERR_ARG, "%s: bad shape", who);
  if ((long long)C * L >= 16777216LL) M2D_FAIL(M2D_ERR_RANGE, "%s: C*L too large", who);
  return M2D_OK;
}

static int bn_apply_bwd(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
                        const float* save_invstd, const float* s_dz, const float* s_dzx, float* dx, int B, int C, int L,
                        int act, float slope, hipStream_t stream);

// the finalisation of the batch statistics folded into a normalisation launch (BnApplyArgs::fin_sums)
struct BnFin {
  const double* sums;
  double count;
  float eps, momentum;
  float* running_mean;
  float* running_var;
};
static bool bn_fin_in_apply() {   // A/B lever (0: the separate m2d_bn_finalize_fwd_kernel launch of rounds 1-5)
  static const bool on = [] { const char* e = getenv("M2D_BN_FIN_IN_APPLY"); return !(e && e[0] == '0'); }();
  return on;
}
static void bn_set_fin(BnApplyArgs& a, const BnFin* fin) {
  if (!fin) return;
  a.fin_sums = fin->sums;
  a.fin_count = fin->count;
  a.fin_eps = fin->eps;
  a.fin_momentum = fin->momentum;
  a.fin_running_mean = fin->running_mean;
  a.fin_running_var = fin->running_var;
}

static int bn_apply_fwd(const float* x, const float* gamma, const float* beta, const float* mean, const float* invstd,
                        float* y, int B, int C, int L, int act, float slope, const float* residual, hipStream_t stream,
                        long long y_batch_stride = 0, const BnFin* fin = nullptr, float* pool_out = nullptr) {
  BnApplyArgs a;
  memset(&a, 0, sizeof(a));
  bn_set_fin(a, fin);
  a.pool_out = pool_out;
  a.out_pitch = y_batch_stride;
  a.x = x;
  a.residual = residual;
  a.gamma = gamma; a.beta = beta;
  a.mean = mean; a.invstd = invstd;
  a.out = y;
  a.C = C; a.L = L; a.L_inv = 1.f / (float)L;
  a.row_len = C * L;
  a.act = act; a.slope = slope;
  M2dProfScope prof(M2D_FAM_BN, stream, 0.0, ((residual ? 3.0 : 2.0) + (pool_out ? 0.5 : 0.0)) * 4.0 * B * C * (double)L,
                    pool_out ? "bn_apply_pool" : "bn_apply", B, C, L);
  return launch_apply(a, B, stream);
}

// The batch statistics as raw sums: sums[2c] = sum x, sums[2c + 1] = sum x^2 over (batch, length), fp64.
// Split from the normalisation so that (a) a producing conv can hand the sums over from its own
// epilogue (m2d_conv1d_fwd's `stats`) and (b) data-parallel ranks can all-reduce them
// (synchronised BatchNorm: global-batch statistics).
int m2d_bn_stats(const float* x, double* sums, int B, int C, int L, void* scratch, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int rc = bn_check("m2d_bn_stats", B, C, L)) return rc;
  BnReduceArgs r;
  memset(&r, 0, sizeof(r));
  r.x = x;
  r.acc = sums;
  if (scratch) {
    bn_use_scratch(r, scratch, C);
    r.fin = 1;
    r.sums_out = sums;
  }
  r.B = B; r.C = C; r.L = L;
  r.mode = 0;
  M2dProfScope prof(M2D_FAM_BN, stream, 0.0, 4.0 * B * C * (double)L, "bn_stats", B, C, L);
  return launch_reduce(r, stream);
}

// Training forward from given sums over `count` elements per channel (count = B*L, or the global
// count under synchronised BatchNorm): mean / invstd, running statistics, then y = residual + act(bn(x)).
// (y_batch_stride: elements between consecutive samples of y - 0 or C * L = dense. A larger stride writes the result
// into a channel block of a wider (B, C', L) buffer: the U-Net's skip concatenations, phase3/archis/default.py:240-245 of
// the reference, are produced in place instead of by a torch.cat pass over both halves.)
int m2d_bn_fwd_sums_to(const float* x, const double* sums, double count, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float* y, float* save_mean, float* save_invstd, int B,
                       int C, int L, float eps, float momentum, int act, float slope, const float* residual,
                       long long y_batch_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int rc = bn_check("m2d_bn_fwd_sums", B, C, L)) return rc;
  if (!sums || !(count > 0.0)) M2D_FAIL(M2D_ERR_ARG, "m2d_bn_fwd_sums: no statistics");
  if (bn_fin_in_apply()) {   // mean / invstd in every thread's prologue, save_* and the running buffers by workgroup (0, 0)
    const BnFin fin = {sums, count, eps, momentum, running_mean, running_var};
    return bn_apply_fwd(x, gamma, beta, save_mean, save_invstd, y, B, C, L, act, slope, residual, stream, y_batch_stride, &fin);
  }
  hipLaunchKernelGGL(m2d_bn_finalize_fwd_kernel, dim3(m2d_ceil_div(C, 256)), dim3(256), 0, stream, sums, save_mean,
                     save_invstd, running_mean, running_var, C, count, eps, momentum);
  M2D_CHECK_LAUNCH("m2d_bn_finalize_fwd_kernel");
  return bn_apply_fwd(x, gamma, beta, save_mean, save_invstd, y, B, C, L, act, slope, residual, stream, y_batch_stride);
}
// The running buffers advanced ONCE MORE with batch statistics they have already been advanced with (same fp64 sums ->
// exactly what a second training-mode forward of the same batch through the same weights would leave behind). The
// reference's loop body runs the generator twice on one batch when a generator iteration follows the critic iteration
// (phase3/train.py:195 and :222): the audio encoder's activations of the second pass equal the first's, so the engine
// reuses them and replays only this update (engine.Phase3Engine, round 5). tmp: 2 C floats of scratch.
int m2d_bn_update_running(const double* sums, double count, float* running_mean, float* running_var, float* tmp, int C,
                          float eps, float momentum, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!sums || !(count > 0.0) || !running_mean || !running_var || !tmp || C <= 0)
    M2D_FAIL(M2D_ERR_ARG, "m2d_bn_update_running: bad arguments");
  hipLaunchKernelGGL(m2d_bn_finalize_fwd_kernel, dim3(m2d_ceil_div(C, 256)), dim3(256), 0, stream, sums, tmp, tmp + C,
                     running_mean, running_var, C, count, eps, momentum);
  M2D_CHECK_LAUNCH("m2d_bn_finalize_fwd_kernel");
  return M2D_OK;
}

int m2d_bn_fwd_sums(const float* x, const double* sums, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float* y, float* save_mean, float* save_invstd, int B,
                    int C, int L, float eps, float momentum, int act, float slope, const float* residual,
                    void* stream_) {
  return m2d_bn_fwd_sums_to(x, sums, count, gamma, beta, running_mean, running_var, y, save_mean, save_invstd, B, C, L, eps,
                            momentum, act, slope, residual, 0, stream_);
}

// The training forward from batch sums (m2d_bn_fwd_sums_to) with the pass that follows it in the U-Net fused in
// (phase3/archis/default.py:235-245 of the reference: MaxPool1d(2, 2) after a skip's BatchNorm, Upsample(x2, linear)
// after a decoder level's):
//   m2d_bn_fwd_sums_pool_to:      y = act(bn(x)) -> y (batch stride as m2d_bn_fwd_sums_to) AND pooled (B, C, L / 2), L even
//   m2d_bn_fwd_sums_upsample2_to: up = upsample2(act(bn(x))) -> (B, C, 2L) at up + b * up_batch_stride; y itself is not
//                                 written. C * L even.
// save_mean / save_invstd and the running buffers as m2d_bn_fwd_sums_to.
int m2d_bn_fwd_sums_pool_to(const float* x, const double* sums, double count, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float* y, float* pooled, float* save_mean,
                            float* save_invstd, int B, int C, int L, float eps, float momentum, int act, float slope,
                            long long y_batch_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int rc = bn_check("m2d_bn_fwd_sums_pool_to", B, C, L)) return rc;
  if (!sums || !(count > 0.0)) M2D_FAIL(M2D_ERR_ARG, "m2d_bn_fwd_sums_pool_to: no statistics");
  if (!pooled || (L & 1)) M2D_FAIL(M2D_ERR_ARG, "m2d_bn_fwd_sums_pool_to: even L and a pooled output");
  const BnFin fin = {sums, count, eps, momentum, running_mean, running_var};
  if (!bn_fin_in_apply()) {
    hipLaunchKernelGGL(m2d_bn_finalize_fwd_kernel, dim3(m2d_ceil_div(C, 256)), dim3(256), 0, stream, sums, save_mean,
                       save_invstd, running_mean, running_var, C, count, eps, momentum);
    M2D_CHECK_LAUNCH("m2d_bn_finalize_fwd_kernel");
  }
  return bn_apply_fwd(x, gamma, beta, save_mean, save_invstd, y, B, C, L, act, slope, nullptr, stream, y_batch_stride,
                      bn_fin_in_apply() ? &fin : nullptr, pooled);
}

int m2d_bn_fwd_sums_upsample2_to(const float* x, const double* sums, double count, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float* up, float* save_mean, float* save_invstd,
                                 int B, int C, int L, float eps, float momentum, int act, float slope,
                                 long long up_batch_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int rc = bn_check("m2d_bn_fwd_sums_upsample2_to", B, C, L)) return rc;
  if (!sums || !(count > 0.0)) M2D_FAIL(M2D_ERR_ARG, "m2d_bn_fwd_sums_upsample2_to: no statistics");
  const long long per = (long long)C * L;
  if (up_batch_stride <= 0) up_batch_stride = 2 * per;
  if ((per & 1) || up_batch_stride < 2 * per || (up_batch_stride & 3) || (((uintptr_t)x | (uintptr_t)up) & 15u))
    M2D_FAIL(M2D_ERR_ARG, "m2d_bn_fwd_sums_upsample2_to: C * L even, 16-byte aligned samples");
  const bool rows = (L & 1) == 0;
  const bool fused_fin = rows && bn_fin_in_apply();   // (the flat form reads the parameters per pair: it keeps the arrays)
  if (!fused_fin) {
    hipLaunchKernelGGL(m2d_bn_finalize_fwd_kernel, dim3(m2d_ceil_div(C, 256)), dim3(256), 0, stream, sums, save_mean,
                       save_invstd, running_mean, running_var, C, count, eps, momentum);
    M2D_CHECK_LAUNCH("m2d_bn_finalize_fwd_kernel");
  }
  BnApplyArgs a;
  memset(&a, 0, sizeof(a));
  const BnFin fin = {sums, count, eps, momentum, running_mean, running_var};
  if (fused_fin) bn_set_fin(a, &fin);
  a.out_pitch = up_batch_stride;
  a.x = x;
  a.gamma = gamma; a.beta = beta;
  a.mean = save_mean; a.invstd = save_invstd;
  a.out = up;
  a.C = C; a.L = L; a.L_inv = 1.f / (float)L;
  a.row_len = C * L;
  a.act = act; a.slope = slope;
  M2dProfScope prof(M2D_FAM_BN, stream, 0.0, 3.0 * 4.0 * B * C * (double)L, "bn_apply_upsample2", B, C, L);
  if (rows) {
    const int rvl = a.row_len / 2;
    unsigned gx, gy;
    if (rvl >= 256) {
      gx = (unsigned)m2d_ceil_div(rvl, 256);
      gy = 4096u / gx;
      if (gy < 1) gy = 1;
      if (gy > (unsigned)B) gy = (unsigned)B;
    } else {
      gx = 1;
      gy = (unsigned)m2d_ceil_div(B, 256 / rvl);
      if (gy > 4096u) gy = 4096u;
    }
    hipLaunchKernelGGL(m2d_bn_upsample2_rows_kernel, dim3(gx, gy), dim3(256), 0, stream, a, B);
    M2D_CHECK_LAUNCH("m2d_bn_upsample2_rows_kernel");
    return M2D_OK;
  }
  const size_t pairs = (size_t)B * (size_t)(per / 2);
  size_t blocks = (pairs + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(m2d_bn_upsample2_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, (unsigned)(per / 2), pairs);
  M2D_CHECK_LAUNCH("m2d_bn_upsample2_flat_kernel");
  return M2D_OK;
}

// Training / eval forward of nn.BatchNorm1d fused with ReLU (act=1) / LeakyReLU (act=2)
// and an optional residual add: y = residual + act(bn(x)).
// training != 0: batch statistics, running stats updated in place (may be NULL),
//                save_mean / save_invstd (C floats each) written for the backward.
// training == 0: running statistics; save_* still written (mean, 1/sqrt(var+eps)).
int m2d_bn_fwd_to(const float* x, const float* gamma, const float* beta, float* running_mean,
                  float* running_var, float* y, float* save_mean, float* save_invstd, int B, int C, int L,
                  float eps, float momentum, int training, int act, float slope, const float* residual,
                  void* ws, size_t ws_bytes, void* scratch, long long y_batch_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int rc = bn_check("m2d_bn_fwd", B, C, L)) return rc;
  if (ws_bytes < m2d_bn_workspace_bytes(C) || !ws) M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_bn_fwd: workspace too small");
  if (training && scratch) {
    // statistics + mean / invstd / running buffers in one launch, then the apply pass
    BnReduceArgs r;
    memset(&r, 0, sizeof(r));
    r.x = x;
    bn_use_scratch(r, scratch, C);
    r.fin = 2;
    r.f0 = save_mean; r.f1 = save_invstd; r.f2 = running_mean; r.f3 = running_mean ? running_var : nullptr;
    r.count = (double)B * L; r.eps = eps; r.momentum = momentum;
    r.B = B; r.C = C; r.L = L;
    r.mode = 0;
    {
      M2dProfScope prof(M2D_FAM_BN, stream, 0.0, 4.0 * B * C * (double)L, "bn_stats", B, C, L);
      if (int rc = launch_reduce(r, stream)) return rc;
    }
    return bn_apply_fwd(x, gamma, beta, save_mean, save_invstd, y, B, C, L, act, slope, residual, stream, y_batch_stride);
  }
  if (training) {
    if (int rc = m2d_bn_stats(x, (double*)ws, B, C, L, nullptr, stream_)) return rc;
    return m2d_bn_fwd_sums_to(x, (const double*)ws, (double)B * L, gamma, beta, running_mean, running_var, y, save_mean,
                              save_invstd, B, C, L, eps, momentum, act, slope, residual, y_batch_stride, stream_);
  }
  if (!running_mean || !running_var) M2D_FAIL(M2D_ERR_ARG, "m2d_bn_fwd: eval mode needs running stats");
  hipLaunchKernelGGL(m2d_bn_eval_stats_kernel, dim3(m2d_ceil_div(C, 256)), dim3(256), 0, stream,
                     (const float*)running_mean, (const float*)running_var, save_mean, save_invstd, C, eps);
  M2D_CHECK_LAUNCH("m2d_bn_eval_stats_kernel");
  return bn_apply_fwd(x, gamma, beta, save_mean, save_invstd, y, B, C, L, act, slope, residual, stream, y_batch_stride);
}
int m2d_bn_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
               float* running_var, float* y, float* save_mean, float* save_invstd, int B, int C, int L,
               float eps, float momentum, int training, int act, float slope, const float* residual,
               void* ws, size_t ws_bytes, void* scratch, void* stream_) {
  return m2d_bn_fwd_to(x, gamma, beta, running_mean, running_var, y, save_mean, save_invstd, B, C, L, eps, momentum, training,
                       act, slope, residual, ws, ws_bytes, scratch, 0, stream_);
}

// Backward reductions as raw sums: sums[2c] = sum dz, sums[2c + 1] = sum dz * xhat, with
// dz = dy * act'(bn(x)) recomputed from x (no y needed).
int m2d_bn_bwd_stats(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
                     const float* save_invstd, double* sums, int B, int C, int L, int act, float slope,
                     void* scratch, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int rc = bn_check("m2d_bn_bwd_stats", B, C, L)) return rc;
  BnReduceArgs r;
  memset(&r, 0, sizeof(r));
  r.x = x; r.dy = dy;
  r.gamma = gamma; r.beta = beta; r.mean = save_mean; r.invstd = save_invstd;
  r.acc = sums;
  if (scratch) {
    bn_use_scratch(r, scratch, C);
    r.fin = 1;
    r.sums_out = sums;
  }
  r.B = B; r.C = C; r.L = L;
  r.mode = 1; r.act = act; r.slope = slope;
  M2dProfScope prof(M2D_FAM_BN, stream, 0.0, 2.0 * 4.0 * B * C * (double)L, "bn_bwd_reduce", B, C, L);
  return launch_reduce(r, stream);
}

// Training-mode backward from the sums: dgamma = sum(dz * xhat), dbeta = sum(dz) from THIS process's
// sums, dx = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)) with the means over the batch the
// statistics were taken over (`sums_global` / `count`: the same buffer and B*L, or all-reduced).
int m2d_bn_bwd_sums(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
                    const float* save_invstd, const double* sums_local, const double* sums_global, double count,
                    float* dx, float* dgamma, float* dbeta, int B, int C, int L, int act, float slope, void* ws,
                    size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int rc = bn_check("m2d_bn_bwd_sums", B, C, L)) return rc;
  if (ws_bytes < m2d_bn_workspace_bytes(C) || !ws) M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_bn_bwd_sums: workspace too small");
  if (!sums_local || !sums_global || !(count > 0.0)) M2D_FAIL(M2D_ERR_ARG, "m2d_bn_bwd_sums: no sums");
  float* s_dz = (float*)((double*)ws + 2 * (size_t)C);
  float* s_dzx = s_dz + C;
  hipLaunchKernelGGL(m2d_bn_finalize_bwd_kernel, dim3(m2d_ceil_div(C, 256)), dim3(256), 0, stream, sums_local,
                     sums_global, dgamma, dbeta, s_dz, s_dzx, C, count);
  M2D_CHECK_LAUNCH("m2d_bn_finalize_bwd_kernel");
  return bn_apply_bwd(dy, x, gamma, beta, save_mean, save_invstd, s_dz, s_dzx, dx, B, C, L, act, slope, stream);
}

static int bn_apply_bwd(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
                        const float* save_invstd, const float* s_dz, const float* s_dzx, float* dx, int B, int C, int L,
                        int act, float slope, hipStream_t stream) {
  BnApplyArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.dy = dy;
  a.gamma = gamma; a.beta = beta; a.mean = save_mean; a.invstd = save_invstd;
  a.s_dz = s_dz; a.s_dzx = s_dzx;
  a.out = dx;
  a.C = C; a.L = L; a.L_inv = 1.f / (float)L;
  a.row_len = C * L;
  a.backward = 1; a.act = act; a.slope = slope;
  M2dProfScope prof(M2D_FAM_BN, stream, 0.0, 3.0 * 4.0 * B * C * (double)L, "bn_bwd_apply", B, C, L);
  return launch_apply(a, B, stream);
}

// Training-mode backward (single process): the two calls above on the workspace.
int m2d_bn_bwd(const float* dy, const float* x, const float* gamma, const float* beta,
               const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta,
               int B, int C, int L, int act, float slope, void* ws, size_t ws_bytes, void* scratch, void* stream_) {
  if (int rc = bn_check("m2d_bn_bwd", B, C, L)) return rc;
  if (ws_bytes < m2d_bn_workspace_bytes(C) || !ws) M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_bn_bwd: workspace too small");
  if (scratch) {
    hipStream_t stream = (hipStream_t)stream_;
    float* s_dz = (float*)((double*)ws + 2 * (size_t)C);
    float* s_dzx = s_dz + C;
    BnReduceArgs r;
    memset(&r, 0, sizeof(r));
    r.x = x; r.dy = dy;
    r.gamma = gamma; r.beta = beta; r.mean = save_mean; r.invstd = save_invstd;
    bn_use_scratch(r, scratch, C);
    r.fin = 3;
    r.f0 = dgamma; r.f1 = dbeta; r.f2 = s_dz; r.f3 = s_dzx;
    r.count = (double)B * L;
    r.B = B; r.C = C; r.L = L;
    r.mode = 1; r.act = act; r.slope = slope;
    {
      M2dProfScope prof(M2D_FAM_BN, stream, 0.0, 2.0 * 4.0 * B * C * (double)L, "bn_bwd_reduce", B, C, L);
      if (int rc = launch_reduce(r, stream)) return rc;
    }
    return bn_apply_bwd(dy, x, gamma, beta, save_mean, save_invstd, s_dz, s_dzx, dx, B, C, L, act, slope, stream);
  }
  if (int rc = m2d_bn_bwd_stats(dy, x, gamma, beta, save_mean, save_invstd, (double*)ws, B, C, L, act, slope, nullptr,
                                stream_))
    return rc;
  return m2d_bn_bwd_sums(dy, x, gamma, beta, save_mean, save_invstd, (const double*)ws, (const double*)ws,
                         (double)B * L, dx, dgamma, dbeta, B, C, L, act, slope, ws, ws_bytes, stream_);
}

// out[c] = sum_{n,l} x[n,c,l] * (mask ? (mask[n,c,l] > 0 ? 1 : slope) : 1)
// Bias gradient of conv1d (C = Cout) and of linear layers (L = 1), with the fused
// activation derivative.
int m2d_channel_sums(const float* x, const float* mask, float slope, float* out, int B, int C, int L,
                     void* ws, size_t ws_bytes, void* scratch, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || C <= 0 || L <= 0) M2D_FAIL(M2D_ERR_ARG, "m2d_channel_sums: bad shape");
  if (!scratch && (ws_bytes < m2d_bn_workspace_bytes(C) || !ws))
    M2D_FAIL(M2D_ERR_WORKSPACE, "m2d_channel_sums: workspace too small");
  BnReduceArgs r;
  memset(&r, 0, sizeof(r));
  r.x = x; r.mask = mask; r.slope = slope;
  r.acc = (double*)ws;
  if (scratch) {
    bn_use_scratch(r, scratch, C);
    r.fin = 4;
    r.f0 = out;
  }
  r.B = B; r.C = C; r.L = L;
  r.mode = 2;
  M2dProfScope prof(M2D_FAM_REDUCE, stream, 0.0, (mask ? 8.0 : 4.0) * B * C * (double)L, "channel_sums", B, C, L);
  int rc = launch_reduce(r, stream);
  if (rc || scratch) return rc;
  hipLaunchKernelGGL(m2d_acc_to_float_kernel, dim3(m2d_ceil_div(C, 256)), dim3(256), 0, stream,
                     (const double*)ws, out, C);
  M2D_CHECK_LAUNCH("m2d_acc_to_float_kernel");
  return M2D_OK;
}

}  // extern "C"
