// Shared plumbing for the m2d HIP library (gfx950 only): error reporting across the
// C-ABI, launch checking, and the optional per-kernel-family event profiler that
// bench.py uses for its roofline line.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define M2D_OK 0
#define M2D_ERR_ARG -1
#define M2D_ERR_HIP -2
#define M2D_ERR_WORKSPACE -3
#define M2D_ERR_RANGE -4

void m2d_set_error(const char* fmt, ...);

#define M2D_FAIL(code, ...)      \
  do {                           \
    m2d_set_error(__VA_ARGS__);  \
    return (code);               \
  } while (0)

#define M2D_CHECK_LAUNCH(name)                                            \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess)                                                \
      M2D_FAIL(M2D_ERR_HIP, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
  } while (0)

// ---- profiler: kernel families ------------------------------------------------
enum M2dFamily {
  M2D_FAM_GEMM = 0,   // implicit-GEMM engine (conv1d fwd/bwd_data/bwd_weight, linear)
  M2D_FAM_BN = 1,     // batch-norm statistics / apply / backward
  M2D_FAM_GRU = 2,    // recurrent step kernels
  M2D_FAM_POINTWISE = 3,
  M2D_FAM_REDUCE = 4,
  M2D_FAM_COUNT = 5
};

struct M2dProfScope {
  int fam;
  hipStream_t stream;
  int slot;
  M2dProfScope(int family, hipStream_t s, double flops, double bytes, const char* tag = nullptr, int d0 = 0,
               int d1 = 0, int d2 = 0);
  ~M2dProfScope();
};

// one tensor of a multi-tensor Adam step: struct M2dAdamItem of include/m2d.h (kept identical)
typedef struct M2dAdamItem {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  long long numel;
  float* pack_fwd;
  float* pack_bwd;
  int cout, cin, ks, reserved;
} M2dAdamItem;

static inline int m2d_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long m2d_ceil_div64(long long a, long long b) { return (a + b - 1) / b; }
