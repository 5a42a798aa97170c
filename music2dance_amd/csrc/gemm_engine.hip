// Implementation of the separable-gather fp32-MFMA GEMM engine (see gemm_engine.h).
#include "gemm_engine.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Wave layout of a BM x BN block tile: NW waves as WM x WN, each owning TM x TN accumulator tiles of 32 x 32.
// NW = 4 (256 threads) everywhere except the LDS-direct 128 x 128 kernel, which also runs with NW = 8 (512 threads, a
// wave owns 64 x 32): a CU holds four 32 KB workgroups (the LDS allocation granule is 1280 B: five do not fit), so
// four-wave workgroups give a SIMD 4 waves to pick from, eight-wave ones 8 at 57-64 VGPRs - what a launch with one or
// two workgroups per CU (the pose critic's 5 GFLOP problems) needs to hide its own operand latency.
template <int BM, int BN, int NW>
struct M2dTiling {
  static constexpr int WM = BM >= 64 ? 2 : 1;
  static constexpr int WN = NW / WM;
  static constexpr int TM = BM / (32 * WM);
  static constexpr int TN = BN / (32 * WN);
  static constexpr int NTH = 64 * NW;
};

// n / d and n % d for 0 <= n < 2^24 (exact in fp32; the launcher rejects larger extents). Branch-free.
__device__ __forceinline__ void m2d_divmod(int n, int d, float inv, int& q, int& r) {
  q = (int)((float)n * inv);
  r = n - q * d;
  const int adj = (r < 0 ? -1 : 0) + (r >= d ? 1 : 0);
  q += adj;
  r -= adj * d;
}

// Workgroup -> tile map (p.tile_map). The hardware deals consecutive workgroup ids round-robin over the 8 XCDs, each
// with an L2 of its own, so with tile = workgroup id (N fastest) the tiles resident on one XCD are ~96 different N
// tiles of ONE M tile: every XCD streams its own copy of that A panel AND a disjoint eighth of B, and an M tile's B
// panel is fetched again for the next M tile thousands of workgroups later. tile_map = 1: workgroups with equal
// id % 8 (they share an XCD: a label, not the XCD's number) take a CONTIGUOUS range of a grouped tile order - groups of
// up to 8 M tiles x all N tiles, M fastest - so the tiles an XCD holds at one time are all M tiles of a few neighbouring
// N tiles (convs: every output-channel tile of the same positions reads the activation tile from that L2 once) or a
// compact 8 x 12 patch of a large GEMM. A pure speed choice: any map is a bijection of the same tiles.
__device__ __forceinline__ void m2d_tile_of(int tile_map, int& bx, int& by) {
  bx = blockIdx.x;
  by = blockIdx.y;
  if (tile_map == 0) return;
  const int nt = gridDim.x, mt = gridDim.y;
  const int T = nt * mt, q = T >> 3, r = T & 7;
  const int lin = by * nt + bx;
  const int xcd = lin & 7, idx = lin >> 3;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int GM = 8;
  const int per = GM * nt;
  const int grp = id / per, rem = id - grp * per;
  const int first = grp * GM;
  const int gsz = mt - first < GM ? mt - first : GM;
  const int nn = rem / gsz;
  by = first + rem - nn * gsz;
  bx = nn;
}

// Staging loads are raw buffer loads: an element that is padding, past a lo tail or past the
// last row gets the byte offset M2D_OOB, which the hardware range check (num_records =
// operand extent < 2^31) turns into 0.0f whatever scalar offset is added. No select on the
// data, no divergent branch, nothing that forces a wait on the load before the LDS store.
#define M2D_OOB 0x80000000u
#define M2D_BAD 0x40000000

__device__ __forceinline__ __amdgpu_buffer_rsrc_t m2d_rsrc(const float* p, unsigned nbytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)0, (int)nbytes, 0x00020000);
}

// voff: per-lane byte offset (range-checked), soff: wave-uniform byte offset (SGPR)
__device__ __forceinline__ float m2d_bload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}

// the same load with the LDS as its destination: lane L of the wave writes dst[L] (M0 = dst, wave-uniform);
// lanes whose offset fails the range check write 0.0f (measured on gfx950), so padding needs no separate store
typedef __attribute__((address_space(3))) float m2d_lds_f;
__device__ __forceinline__ void m2d_bload_lds(__amdgpu_buffer_rsrc_t r, float* dst, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (m2d_lds_f*)dst, 4, (int)voff, soff, 0, 0);
}

// element offset of row-side index hi (two-level for window views, see M2dOperand)
__device__ __forceinline__ int m2d_hi_offset(const M2dOperand& op, int hi) {
  if (op.rdiv2 <= 0) return hi * op.r_hi_stride;
  int q, r;
  m2d_divmod(hi, op.rdiv2, op.rdiv2_inv, q, r);
  return q * op.r_hi2_stride + r * op.r_hi_stride;
}

// Per-thread bookkeeping for one operand tile of BR rows x M2D_BK k-slots of a chunk
// (hi, lo0 .. lo0 + 15).
//   k-fast map  : thread owns ONE k-slot (tid % BK) and NE rows (tid / BK + i * 256 / BK):
//                 operands whose lo index is the contiguous one (packed weights, dy in
//                 bwd_weight, plain row-major matrices). No window test (launcher enforces).
//                 address = eoff[i] (row + slot, per thread) + S (chunk scalar, in soffset)
//   row-fast map: thread owns ONE row (tid % BR) and NE consecutive k-slots from
//                 kb = (tid / BR) * NE: address = eoff (row + kb) + S (chunk scalar, added on
//                 the vector side so that S may be negative) + i * k_lo_stride (loop-invariant
//                 scalar, in soffset for uniform chunks). The hardware range check covers the
//                 vector offset only, so that part is always a real element's offset.
// UNIFORM chunks (no lo tail, window independent of lo or known to pass): validity is one test
// per thread per chunk. General chunks test every element.
template <bool KF, int BR, bool MASKED, int NTH = 256>
struct TileMap {
  static constexpr int NE = BR * M2D_BK / NTH;
  static constexpr int NR = KF ? NE : 1;
  static constexpr int NM = MASKED ? NE : 1;
  unsigned eoff[NR];
  int posr;  // row-fast: window position of the row incl. kb * k_pos_lo; M2D_BAD for rows past the extent
  int kb;    // k-fast: the thread's slot; row-fast: its first slot
  unsigned lim_eff;
  bool force_one;  // row-fast: this thread's row is the operand's all-ones row
  float one_val;   // what that row reads as in the staged chunk (1, or 0 for chunks below ones_from_hi)
  float v[NE];   // staged values
  float mv[NM];  // staged mask values (MASKED only)

  __device__ __forceinline__ void prep(const M2dOperand& op, int row0, int tid) {
    lim_eff = op.lim > 0 ? (unsigned)op.lim : (unsigned)(M2D_BAD - 1);
    force_one = false;
    one_val = 1.f;
    if constexpr (KF) {
      kb = tid % M2D_BK;
      posr = 0;
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int g = row0 + tid / M2D_BK + i * (NTH / M2D_BK);
        const bool rv = g < op.nrows;
        int hi, lo;
        m2d_divmod(rv ? g : 0, op.rdiv, op.rdiv_inv, hi, lo);
        const int off = m2d_hi_offset(op, hi) + lo * op.r_lo_stride + op.r_off + kb * op.k_lo_stride;
        eoff[i] = rv ? ((unsigned)off << 2) : M2D_OOB;
      }
    } else {
      const int kb_t = (tid / BR) * NE;
      kb = BR >= 64 ? __builtin_amdgcn_readfirstlane(kb_t) : kb_t;  // wave-uniform when a wave spans <= 1 k-group
      const int g = row0 + tid % BR;
      const bool rv = g < op.nrows;
      force_one = (g + 1 == op.ones_row_p1);
      int hi, lo;
      m2d_divmod(rv ? g : 0, op.rdiv, op.rdiv_inv, hi, lo);
      const int off = m2d_hi_offset(op, hi) + lo * op.r_lo_stride + op.r_off + kb * op.k_lo_stride;
      eoff[0] = (unsigned)off << 2;
      posr = rv ? (op.lim > 0 ? lo * op.r_pos_mul + op.r_pos_off + kb * op.k_pos_lo : 0) : M2D_BAD;
    }
  }

  // LDS-direct staging (row-fast): a wave must fill lane-consecutive dwords of the [k][row] image. BR >= 64: as prep().
  // BR == 32: the two half-waves take ADJACENT k-slots (slot = 4 * wave + (lane >> 5) + 2 i), so that lanes 0..63 of
  // one load cover two whole 32-row lines of the image.
  static constexpr int DL_KSTEP = BR >= 64 ? 1 : 64 / BR;
  __device__ __forceinline__ void prep_dl(const M2dOperand& op, int row0, int tid) {
    static_assert(!KF && !MASKED && (BR == 32 || BR >= 64), "LDS-direct staging: row-fast, unmasked");
    static_assert(BR >= 64 || NTH == 256, "32-row LDS-direct staging: four waves");
    if constexpr (BR >= 64) {
      prep(op, row0, tid);
    } else {
      lim_eff = op.lim > 0 ? (unsigned)op.lim : (unsigned)(M2D_BAD - 1);
      force_one = false;
      kb = 4 * (tid >> 6) + ((tid >> 5) & 1);
      const int g = row0 + (tid & (BR - 1));
      const bool rv = g < op.nrows;
      int hi, lo;
      m2d_divmod(rv ? g : 0, op.rdiv, op.rdiv_inv, hi, lo);
      const int off = m2d_hi_offset(op, hi) + lo * op.r_lo_stride + op.r_off + kb * op.k_lo_stride;
      eoff[0] = (unsigned)off << 2;
      posr = rv ? (op.lim > 0 ? lo * op.r_pos_mul + op.r_pos_off + kb * op.k_pos_lo : 0) : M2D_BAD;
    }
  }

  __device__ __forceinline__ void fetch(__amdgpu_buffer_rsrc_t rs, __amdgpu_buffer_rsrc_t rm, int i, unsigned voff,
                                        int soff) {
    v[i] = m2d_bload(rs, voff, soff);
    if constexpr (MASKED) mv[i] = m2d_bload(rm, voff, soff);  // rm has 0 records when the operand has no mask
  }

  // stage chunk (hi, lo0) into registers
  template <bool UNIFORM>
  __device__ __forceinline__ void load(const M2dOperand& op, __amdgpu_buffer_rsrc_t rs, __amdgpu_buffer_rsrc_t rm,
                                       int hi, int lo0, int kdiv) {
    int hoff = hi * op.k_hi_stride;
    if (op.kdiv2 > 0) {  // window view on the k side: hi = b * T + t (one scalar division per chunk)
      const int q = hi / op.kdiv2;
      hoff = q * op.k_hi2_stride + (hi - q * op.kdiv2) * op.k_hi_stride;
    }
    const int S = (hoff + lo0 * op.k_lo_stride) << 2;  // wave-uniform
    if constexpr (!KF) one_val = hi >= op.ones_from_hi ? 1.f : 0.f;
    if constexpr (KF) {
      if constexpr (UNIFORM) {
#pragma unroll
        for (int i = 0; i < NE; ++i) fetch(rs, rm, i, eoff[i], S);
      } else {
        const bool tv = lo0 + kb < kdiv;
#pragma unroll
        for (int i = 0; i < NE; ++i) fetch(rs, rm, i, tv ? eoff[i] : M2D_OOB, S);
      }
    } else {
      const int P = hi * op.k_pos_hi + lo0 * op.k_pos_lo;  // wave-uniform
      const unsigned full = eoff[0] + (unsigned)S;
      const int ls4 = op.k_lo_stride << 2;
      if constexpr (UNIFORM) {
        const unsigned voff = ((unsigned)(posr + P) < lim_eff) ? full : M2D_OOB;
#pragma unroll
        for (int i = 0; i < NE; ++i) fetch(rs, rm, i, voff, i * ls4);
      } else {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
          // the range check sees the vector offset only: it must be the element's own offset
          // (the thread's first slot may be padding while slot i is not)
          const int pp = (lo0 + kb + i < kdiv) ? P + i * op.k_pos_lo : M2D_BAD;
          fetch(rs, rm, i, ((unsigned)(posr + pp) < lim_eff) ? full + (unsigned)(i * ls4) : M2D_OOB, 0);
        }
      }
    }
  }

  // row-fast map, unmasked, no all-ones row: stage chunk (hi, lo0) STRAIGHT into the LDS image [k][row] (leading
  // dimension LD): a wave's 64 rows of one k-slot are lane-consecutive dwords. `img` = the operand's image of
  // the stage being filled; no staging registers, no ds_write.
  template <bool UNIFORM, int LD>
  __device__ __forceinline__ void load_lds(const M2dOperand& op, __amdgpu_buffer_rsrc_t rs, int hi, int lo0, int kdiv,
                                           float* img, int tid) const {
    static_assert(!KF && !MASKED, "LDS-direct staging: row-fast, unmasked");
    constexpr int KS = DL_KSTEP;
    int hoff = hi * op.k_hi_stride;
    const int S = (hoff + lo0 * op.k_lo_stride) << 2;
    const int P = hi * op.k_pos_hi + lo0 * op.k_pos_lo;
    const unsigned full = eoff[0] + (unsigned)S;
    const int ls4 = (op.k_lo_stride * KS) << 2;
    // wave-uniform destination: the image line of lane 0's slot, at the wave's first row
    const int kb0 = BR >= 64 ? kb : 4 * (__builtin_amdgcn_readfirstlane(tid) >> 6);
    const int row0 = BR >= 64 ? (__builtin_amdgcn_readfirstlane(tid) & (BR - 1) & ~63) : 0;
    float* dst = img + kb0 * LD + row0;
    if constexpr (UNIFORM) {
#ifdef M2D_X_NO_WINDOW  // experiment: no window test in uniform chunks
      const unsigned voff = full;
#else
      const unsigned voff = ((unsigned)(posr + P) < lim_eff) ? full : M2D_OOB;
#endif
#pragma unroll
      for (int i = 0; i < NE; ++i) m2d_bload_lds(rs, dst + (KS * i) * LD, voff, i * ls4);
    } else {
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int pp = (lo0 + kb + KS * i < kdiv) ? P + (KS * i) * op.k_pos_lo : M2D_BAD;
        m2d_bload_lds(rs, dst + (KS * i) * LD, ((unsigned)(posr + pp) < lim_eff) ? full + (unsigned)(i * ls4) : M2D_OOB, 0);
      }
    }
  }

  // LDS image is [k][row] with leading dimension LD (LD % 32 == 2 keeps the k-fast
  // writes conflict-free; row-fast writes and fragment reads are conflict-free anyway).
  template <int LD>
  __device__ __forceinline__ void store(const M2dOperand& op, float* s, int tid) const {
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int kl = KF ? (tid % M2D_BK) : ((tid / BR) * NE + i);
      const int rl = KF ? (tid / M2D_BK + i * (NTH / M2D_BK)) : (tid % BR);
      float x = v[i];
      if constexpr (MASKED) x *= (mv[i] > 0.f ? 1.f : op.mask_slope);
      if constexpr (!KF) x = force_one ? one_val : x;
      s[kl * LD + rl] = x;
    }
  }
};

// wave-uniform chunk cursor: chunk (hi, lo0); a chunk is UNIFORM when it holds no lo tail and
// both operands' windows are lo-independent there. Conv layers walk lo blocks in the outer loop
// (16 channels, then all their taps): the activation lines a tile touches for one channel block
// stay in L1 across the taps.
struct ChunkCursor {
  int hi, lo0, kdiv, nhi;
  int lo_outer;                // chunk order: 0 = hi outer / lo inner, 1 = lo block outer / hi inner
  int a_lo, a_hi, b_lo, b_hi;  // safe lo ranges of the two operands
  __device__ __forceinline__ void seek(int c, int cph) {
    if (lo_outer) {
      const int blk = c / nhi;
      hi = c - blk * nhi;
      lo0 = blk * M2D_BK;
    } else {
      hi = c / cph;
      lo0 = (c - hi * cph) * M2D_BK;
    }
  }
  __device__ __forceinline__ bool past() const { return lo_outer ? lo0 >= kdiv : hi >= nhi; }
  __device__ __forceinline__ bool uniform() const {
    return (!past()) & (lo0 + M2D_BK <= kdiv) & (lo0 >= a_lo) & (lo0 + M2D_BK <= a_hi) & (lo0 >= b_lo) &
           (lo0 + M2D_BK <= b_hi);
  }
  // lo extent for the general path: 0 past the end (the chunk after the last one loads zeros)
  __device__ __forceinline__ int extent() const { return past() ? 0 : kdiv; }
  __device__ __forceinline__ void next() {
    if (lo_outer) {
      hi += 1;
      const bool wrap = hi >= nhi;
      hi = wrap ? 0 : hi;
      lo0 += wrap ? M2D_BK : 0;
    } else {
      lo0 += M2D_BK;
      const bool wrap = lo0 >= kdiv;
      lo0 = wrap ? 0 : lo0;
      hi += wrap ? 1 : 0;
    }
  }
};

// byte offset of an element's mask value (M2dOutMap.mask_wrap; wrapb = 0xffffffff: the mask is as large as the output)
__device__ __forceinline__ unsigned m2d_mask_off(unsigned voff, unsigned wrapb) {
  return (voff != M2D_OOB && voff >= wrapb) ? voff - wrapb : voff;
}

// -> the value stored at out[addr]; with O.sum_out the second output is written here too
__device__ __forceinline__ float m2d_epilogue(const M2dOutMap& o, float v, int row, int col, int addr) {
  const int maddr = (o.mask_wrap && addr >= (int)o.mask_wrap) ? addr - (int)o.mask_wrap : addr;
  if (o.bias_mode == 1) v += o.bias[row];
  else if (o.bias_mode == 2) v += o.bias[col];
  if (o.act == 1) v = v > 0.f ? v : 0.f;
  else if (o.act == 2) v = v > 0.f ? v : v * o.slope;
  if (o.mask_last) {
    if (o.residual) v += o.residual[addr];
    if (o.mask) v *= (o.mask[maddr] > 0.f ? 1.f : o.mask_slope);
    return v;
  }
  if (o.mask) v *= (o.mask[maddr] > 0.f ? 1.f : o.mask_slope);
  if (o.sum_out) {
    o.sum_out[addr] = v + o.residual[addr];
    return v;
  }
  if (o.residual) v += o.residual[addr];
  return v;
}

// Multiply chunk `cur` out of LDS and write the staged registers (next chunk, loads issued by
// the caller) into the other buffer.
template <int BM, int BN, bool AKF, bool BKF, bool MASKED>
__device__ __forceinline__ void m2d_chunk_mma(const TileMap<AKF, BM, MASKED>& ta, const TileMap<BKF, BN, MASKED>& tb,
                                              const M2dOperand& A, const M2dOperand& B, float* smem, int cur,
                                              int tid, int wm, int wn, int l31, int lh,
                                              f32x16 (&acc)[BM / (32 * (BM >= 64 ? 2 : 1))][BN / (32 * (4 / (BM >= 64 ? 2 : 1)))]) {
  constexpr int LDA = BM + M2D_LDPAD;
  constexpr int LDB = BN + M2D_LDPAD;
  constexpr int WM = BM >= 64 ? 2 : 1;
  constexpr int WN = 4 / WM;
  constexpr int TM = BM / (32 * WM);
  constexpr int TN = BN / (32 * WN);
  constexpr int STAGE = M2D_BK * (LDA + LDB);
  const float* as = smem + cur * STAGE + wm * (TM * 32) + l31;
  const float* bs = smem + cur * STAGE + M2D_BK * LDA + wn * (TN * 32) + l31;
  float fa[M2D_BK / 2][TM], fb[M2D_BK / 2][TN];
#pragma unroll
  for (int kk = 0; kk < M2D_BK / 2; ++kk) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[kk][i] = as[(2 * kk + lh) * LDA + i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[kk][j] = bs[(2 * kk + lh) * LDB + j * 32];
  }
#pragma unroll
  for (int kk = 0; kk < M2D_BK / 2; ++kk)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
  float* nxt = smem + (cur ^ 1) * STAGE;
  ta.template store<LDA>(A, nxt, tid);
  tb.template store<LDB>(B, nxt + M2D_BK * LDA, tid);
  // software pipeline: the fragments of k-step kk+2 are read under the MFMAs of k-step kk
  __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
  for (int kk = 0; kk < M2D_BK / 2; ++kk) {
    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
    if (kk < M2D_BK / 2 - 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
  }
}

// Accumulator start values: the bias (per row or per column) instead of zero - the MFMA chain then carries it and the
// epilogue has no bias lookup at all (bias[row] was a global load on every row's dependency chain there). Not under the
// two-launch split-K (its reduction kernel adds the bias); with the in-kernel fix-up only split 0 starts from the bias
// (the last arriver sums every split's image, split 0's included, from zero).
template <int BM, int BN, int NW = 4>
__device__ __forceinline__ void m2d_acc_init(const M2dGemmParams& p, const M2dOutMap& O, int N, int split, int m0, int n0, int wm,
                                             int wn, int l31, int lh,
                                             f32x16 (&acc)[M2dTiling<BM, BN, NW>::TM][M2dTiling<BM, BN, NW>::TN]) {
  constexpr int TM = M2dTiling<BM, BN, NW>::TM;
  constexpr int TN = M2dTiling<BM, BN, NW>::TN;
  const bool with_bias = O.bias_mode != 0 && (p.splits <= 1 || (p.tickets && split == 0));
  if (!with_bias) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    return;
  }
  if (O.bias_mode == 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float b = row < p.M ? O.bias[row] : 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j][r] = b;
      }
  } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (TN * 32) + j * 32 + l31;
      const float b = col < N ? O.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = b;
    }
  }
}

// Tile epilogue shared by the staging variants. C/D layout of the 32x32 MFMA: col = lane & 31,
//      row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
// Every 32x32 accumulator tile goes through a wave-private 4 KB LDS image [row][32 columns] (the stage buffers are free
// after the last chunk) and is written out by a rolled loop over the image:
//   WIDE  (launcher-checked: unit column stride, pitches / offsets / N multiples of 4, no window / redirect / row map):
//         a lane takes 4 consecutive columns of one row - 16-byte stores, 8 rows x 128 B per instruction;
//   else  a lane takes one element - column = lane & 31, two rows per step - through the full output map (column
//         divmod, window, sub-pixel row map, redirect column).
// Round 4 rewrite. Phase stamps (tools/phase_stamps.py, -DM2D_STAMP builds) showed a workgroup spending 10-55 us in its
// epilogue - a tenth of a 300-600 us conv launch. Cause: a global load (bias[row], the output mask, the residual) in
// front of every store. `s_waitcnt vmcnt` counts loads AND stores on this part, in issue order, so waiting for item
// q's load also waits for item q - 1's store to be acknowledged: 16-64 store round trips (1-3 us each under load) in
// series per wave. Now NO load sits between two stores: the bias rides in the accumulators (m2d_acc_init), the output
// mask of a whole band is fetched in one burst right after the dump and kept as ONE BIT per element (the multiplier is
// 1 or mask_slope), and the pass over the image only reads LDS and stores. (The residual, rare, is still looked up in
// the pass - inside its own uniform branch, so that launches without one carry no wait.)
__device__ __forceinline__ float4 m2d_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void m2d_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

typedef unsigned int m2d_u32x4 __attribute__((__vector_size__(16)));
typedef float m2d_vf32x4 __attribute__((__vector_size__(16)));
__device__ __forceinline__ void m2d_bstore1(__amdgpu_buffer_rsrc_t r, unsigned voff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, 0, 0);
}
__device__ __forceinline__ void m2d_bstore4(__amdgpu_buffer_rsrc_t r, unsigned voff, float4 v) {
  m2d_vf32x4 w;
  w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(m2d_u32x4, w), r, (int)voff, 0, 0);
}
__device__ __forceinline__ float4 m2d_bload4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  const m2d_vf32x4 w = __builtin_bit_cast(m2d_vf32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
  return make_float4(w[0], w[1], w[2], w[3]);
}

// All global accesses of the pass are raw buffer operations with a 32-bit byte offset (the launcher checks that the
// output map stays below 2 GiB): one address register per access, and an element that must not be touched simply gets
// the offset M2D_OOB - the range check drops the store / returns 0.0 - instead of a branch around the instruction.
template <int BM, int BN, bool WIDE, int NW = 4>
__device__ __forceinline__ void m2d_tile_epilogue(const M2dGemmParams& p, const M2dOutMap& O, int N, int split, int m0,
                                                  int n0, int wm, int wn, int lane, float* wl,
                                                  f32x16 (&acc)[M2dTiling<BM, BN, NW>::TM][M2dTiling<BM, BN, NW>::TN]) {
  constexpr int WN = M2dTiling<BM, BN, NW>::WN;
  constexpr int TM = M2dTiling<BM, BN, NW>::TM;
  constexpr int TN = M2dTiling<BM, BN, NW>::TN;
  const int l31 = lane & 31, lh = lane >> 5;
  // two-launch split-K: the raw partial tile goes to this split's slab [M][N] - the same pass under a plain map
  const bool slab = p.splits > 1 && !p.tickets;
  const bool stats = O.row_part != nullptr && !slab;
  const bool has_mask = O.mask != nullptr && !slab, has_res = O.residual != nullptr && !slab;
  const bool two_out = O.sum_out != nullptr && !slab;
  const int act = slab ? 0 : O.act;
  const float ms = O.mask_slope;
  const int m_stride = slab ? p.N : O.m_stride;
  const int m_div = slab ? 0 : O.m_div;
  const int c_lim = slab ? 0 : O.c_lim;
  const int redirect = slab ? 0 : O.redirect_col_p1;
  const float* out_base = slab ? p.slab + (size_t)split * p.M * p.N : O.out;
  // (descriptors are built where they are used: four of them held across the pass spill scalar registers)
#define M2D_RS(ptr) m2d_rsrc((ptr), 0x7ffffffcu)
  // BM = 128: both tiles of a 32-row band are dumped together (2 x 4 KB per wave = the whole 32 KB of stage buffers)
  // (only the 16-byte pass needs the second tile's registers; the one-element pass takes a tile at a time)
  constexpr int NT = (BM == 128 && WIDE) ? TN : 1;   // (TN = 1 with eight waves: a tile at a time)
  // WIDE: lane -> row rl0 + 8 it, columns c4 .. c4 + 3 (it = 0..3); else: lane -> row 2 it + lh, column l31 (it = 0..15)
  const int rl0 = lane >> 3, c4 = (lane & 7) * 4;
  constexpr int PER_TILE = WIDE ? 4 : 16;
  constexpr int ITEMS = NT * PER_TILE;           // per band
  // items whose mask is fetched in one burst (one bit each fits 32 bits); 128-row tiles: 8 - sixteen loads in flight
  // beside 48 live accumulators would cost the LDS-direct kernel its fifth wave per SIMD
  constexpr int GROUP = WIDE ? ITEMS : (BM == 128 ? 8 : 16);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int rowb = m0 + wm * (TM * 32) + i * 32;
#pragma unroll
    for (int j0 = 0; j0 < TN; j0 += NT) {
#pragma unroll
      for (int jj = 0; jj < NT; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) wl[jj * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = acc[i][j0 + jj][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);  // (nothing of the pass above the dump: its registers are the ones the dump frees)
      // per-tile column state of this lane (NT <= 2 tiles: picked by select)
      int caddr_t[NT], pos_t[NT];
      bool cok_t[NT], red_t[NT];
#pragma unroll
      for (int jj = 0; jj < NT; ++jj) {
        const int col = n0 + wn * (TN * 32) + (j0 + jj) * 32 + (WIDE ? c4 : l31);
        const bool cv = col < N;
        if (slab) {
          caddr_t[jj] = col;
          pos_t[jj] = 0;
        } else {
          int chi, clo;
          m2d_divmod(cv ? col : 0, O.cdiv, O.cdiv_inv, chi, clo);
          caddr_t[jj] = chi * O.c_hi_stride + clo * (WIDE ? 1 : O.c_lo_stride) + O.c_off;
          pos_t[jj] = clo * O.c_pos_mul + O.c_pos_off;
        }
        red_t[jj] = !WIDE && cv && (col + 1 == redirect);
        bool cok = cv;
        if (!WIDE && c_lim > 0 && m_div <= 0) cok = cok && ((unsigned)pos_t[jj] < (unsigned)c_lim);
        cok_t[jj] = cok;
      }
      // ---- fast pass: no residual, plain row map (and, one element per lane: no statistics, no redirect column in
      // this tile). A dozen instructions per element, no branch: the general pass below spends ~150 on its uniform
      // feature tests, and a lane has 64 elements (20 us per workgroup on a plain GEMM tile, measured).
      //   activation: max(x, 0) + s min(x, 0) with s = 1 / 0 / slope (none / ReLU / leaky) - equal to the selects up to
      //   the sign of a zero; mask: one bit per element, fetched in one burst per tile
      {
        const float act_s = act == 0 ? 1.f : (act == 1 ? 0.f : O.slope);
        bool fast = !has_res && m_div <= 0;
        // (one element per lane: no redirect column in this tile; statistics only in the form that re-reads the image in
        // 16-byte pieces below - no output mask, no position window)
        if constexpr (!WIDE) fast = fast && (!stats || (!has_mask && c_lim <= 0 && (p.stats_narrow_fast != 0))) && __ballot(red_t[0]) == 0ull;
        if (fast) {
          if constexpr (!WIDE) {
            if (stats) {
              // Row statistics of a one-element-per-lane tile (output rows that are not 16-byte aligned: WaveGAN's 193- and
              // 794-position rows): the general pass sums a row over its 32 lanes with five shuffle steps per ELEMENT (160
              // per tile and lane, round 5: those launches ran at 93 TFLOP/s against 110-118 for aligned rows). The image
              // is still in LDS: read it once more the way the 16-byte pass does (lane -> row rl0 + 8 it, four columns),
              // three shuffle steps per ROW PIECE. Same values as stored: activation applied, columns past N dropped.
#pragma unroll
              for (int jj = 0; jj < NT; ++jj) {
                const int colb = n0 + wn * (TN * 32) + (j0 + jj) * 32 + c4;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                  const int rl = rl0 + 8 * it;
                  float4 x = m2d_ld4(wl + jj * 1024 + rl * 32 + c4);
                  x.x = colb + 0 < N ? fmaxf(x.x, 0.f) + act_s * fminf(x.x, 0.f) : 0.f;
                  x.y = colb + 1 < N ? fmaxf(x.y, 0.f) + act_s * fminf(x.y, 0.f) : 0.f;
                  x.z = colb + 2 < N ? fmaxf(x.z, 0.f) + act_s * fminf(x.z, 0.f) : 0.f;
                  x.w = colb + 3 < N ? fmaxf(x.w, 0.f) + act_s * fminf(x.w, 0.f) : 0.f;
                  float a1 = (x.x + x.y) + (x.z + x.w);
                  float a2 = (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
#pragma unroll
                  for (int off = 4; off > 0; off >>= 1) {
                    a1 += __shfl_xor(a1, off, 64);
                    a2 += __shfl_xor(a2, off, 64);
                  }
                  if ((lane & 7) == 0 && rowb + rl < p.M) {
                    float* dst = O.row_part + ((size_t)(((n0 / BN) * WN + wn) * TN + j0 + jj) * p.M + rowb + rl) * 2;
                    dst[0] = a1;
                    dst[1] = a2;
                  }
                }
              }
            }
          }
          const __amdgpu_buffer_rsrc_t rso = m2d_rsrc(out_base, 0x7ffffffcu);
#pragma unroll
          for (int jj = 0; jj < NT; ++jj) {
            constexpr int NI = PER_TILE;  // items of one tile: rows r0 + RS it
            constexpr int RS = WIDE ? 8 : 2;
            const int r0 = WIDE ? rl0 : lh;
            const unsigned step = (unsigned)(RS * m_stride) << 2;
            const unsigned v0 = cok_t[jj] ? (unsigned)((rowb + r0) * m_stride + caddr_t[jj]) << 2 : M2D_OOB;
            const int nrow = p.M - rowb - r0;  // item `it` is inside the matrix iff RS it < nrow
            unsigned keep = 0xffffffffu;
            if (has_mask) {
              const __amdgpu_buffer_rsrc_t rsm = m2d_rsrc(O.mask, 0x7ffffffcu);
              const unsigned wrapb = O.mask_wrap ? O.mask_wrap * 4u : 0xffffffffu;   // (formed where it is used: register budget)
              keep = 0u;
              if constexpr (WIDE) {
                float4 mraw[NI];
#pragma unroll
                for (int it = 0; it < NI; ++it)
                  mraw[it] = m2d_bload4(rsm, m2d_mask_off((RS * it < nrow && v0 != M2D_OOB) ? v0 + (unsigned)it * step : M2D_OOB, wrapb));
#pragma unroll
                for (int it = 0; it < NI; ++it)
                  keep |= ((mraw[it].x > 0.f ? 1u : 0u) | (mraw[it].y > 0.f ? 2u : 0u) | (mraw[it].z > 0.f ? 4u : 0u) | (mraw[it].w > 0.f ? 8u : 0u)) << (4 * it);
              } else {
                float mraw[NI];
#pragma unroll
                for (int it = 0; it < NI; ++it)
                  mraw[it] = m2d_bload(rsm, m2d_mask_off((RS * it < nrow && v0 != M2D_OOB) ? v0 + (unsigned)it * step : M2D_OOB, wrapb), 0);
#pragma unroll
                for (int it = 0; it < NI; ++it) keep |= (mraw[it] > 0.f ? 1u : 0u) << it;
              }
            }
            const float* img = wl + jj * 1024 + r0 * 32 + (WIDE ? c4 : l31);
#pragma unroll
            for (int it = 0; it < NI; ++it) {
              const bool ok = RS * it < nrow && v0 != M2D_OOB;
              const unsigned voff = ok ? v0 + (unsigned)it * step : M2D_OOB;
              if constexpr (WIDE) {
                float4 x = m2d_ld4(img + it * (RS * 32));
                const unsigned kb = keep >> (4 * it);
                x.x = (fmaxf(x.x, 0.f) + act_s * fminf(x.x, 0.f)) * ((kb & 1u) ? 1.f : ms);
                x.y = (fmaxf(x.y, 0.f) + act_s * fminf(x.y, 0.f)) * ((kb & 2u) ? 1.f : ms);
                x.z = (fmaxf(x.z, 0.f) + act_s * fminf(x.z, 0.f)) * ((kb & 4u) ? 1.f : ms);
                x.w = (fmaxf(x.w, 0.f) + act_s * fminf(x.w, 0.f)) * ((kb & 8u) ? 1.f : ms);
                m2d_bstore4(rso, voff, x);
                if (stats) {  // this tile's 32 columns of the row: the 8 lanes that share it
                  float a1 = ok ? (x.x + x.y) + (x.z + x.w) : 0.f;
                  float a2 = ok ? (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w) : 0.f;
#pragma unroll
                  for (int off = 4; off > 0; off >>= 1) {
                    a1 += __shfl_xor(a1, off, 64);
                    a2 += __shfl_xor(a2, off, 64);
                  }
                  if ((lane & 7) == 0 && RS * it < nrow) {
                    float* dst = O.row_part + ((size_t)(((n0 / BN) * WN + wn) * TN + j0 + jj) * p.M + rowb + r0 + RS * it) * 2;
                    dst[0] = a1;
                    dst[1] = a2;
                  }
                }
              } else {
                float x = img[it * (RS * 32)];
                x = (fmaxf(x, 0.f) + act_s * fminf(x, 0.f)) * (((keep >> it) & 1u) ? 1.f : ms);
                m2d_bstore1(rso, voff, x);
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
          continue;
        }
      }
      // ---- general pass
      // item q of the dumped tiles -> (tile jj, image row, output row, byte offset of the element or M2D_OOB)
      auto locate = [&](int q, int& jj, int& rl, int& row, unsigned& voff) {
        jj = NT > 1 ? q / PER_TILE : 0;
        const int it = q - jj * PER_TILE;
        rl = WIDE ? rl0 + 8 * it : 2 * it + lh;
        row = rowb + rl;
        const int caddr = NT > 1 && jj ? caddr_t[NT - 1] : caddr_t[0];
        bool ok = row < p.M && (NT > 1 && jj ? cok_t[NT - 1] : cok_t[0]) && !(NT > 1 && jj ? red_t[NT - 1] : red_t[0]);
        int addr;
        if (!WIDE && m_div > 0) {  // sub-pixel row map (the quad epilogue's fall-back)
          int mhi, mlo;
          m2d_divmod(row, m_div, 1.f / (float)m_div, mhi, mlo);
          const int pos = NT > 1 && jj ? pos_t[NT - 1] : pos_t[0];
          if (c_lim > 0) ok = ok && ((unsigned)(pos + mlo * O.m_pos_mul) < (unsigned)c_lim);
          addr = mhi * m_stride + mlo * O.m_lo_stride + caddr;
        } else {
          addr = row * m_stride + caddr;
        }
        voff = ok ? (unsigned)addr << 2 : M2D_OOB;
      };
#pragma unroll 1
      for (int q0 = 0; q0 < ITEMS; q0 += GROUP) {
        // the group's output mask in ONE burst - no load between two stores of the group - kept as one bit per element
        unsigned keep = 0xffffffffu;
        if (has_mask) {
          keep = 0u;
          const unsigned wrapb = O.mask_wrap ? O.mask_wrap * 4u : 0xffffffffu;
          if constexpr (WIDE) {
            float4 mraw[GROUP];
#pragma unroll
            for (int q = 0; q < GROUP; ++q) {
              int jj, rl, row;
              unsigned voff;
              locate(q0 + q, jj, rl, row, voff);
              mraw[q] = m2d_bload4(M2D_RS(O.mask), m2d_mask_off(voff, wrapb));
            }
#pragma unroll
            for (int q = 0; q < GROUP; ++q)
              keep |= ((mraw[q].x > 0.f ? 1u : 0u) | (mraw[q].y > 0.f ? 2u : 0u) | (mraw[q].z > 0.f ? 4u : 0u) | (mraw[q].w > 0.f ? 8u : 0u)) << (4 * q);
          } else {
            float mraw[GROUP];
#pragma unroll
            for (int q = 0; q < GROUP; ++q) {
              int jj, rl, row;
              unsigned voff;
              locate(q0 + q, jj, rl, row, voff);
              mraw[q] = m2d_bload(M2D_RS(O.mask), m2d_mask_off(voff, wrapb), 0);
            }
#pragma unroll
            for (int q = 0; q < GROUP; ++q) keep |= (mraw[q] > 0.f ? 1u : 0u) << q;
          }
        }
#pragma unroll 2
        for (int qq = 0; qq < GROUP; ++qq) {
          int jj, rl, row;
          unsigned voff;
          locate(q0 + qq, jj, rl, row, voff);
          const bool ok = voff != M2D_OOB;
          float a1 = 0.f, a2 = 0.f;
          if constexpr (WIDE) {
            float4 x = m2d_ld4(wl + jj * 1024 + rl * 32 + c4);
            if (act == 1) {
              x.x = x.x > 0.f ? x.x : 0.f; x.y = x.y > 0.f ? x.y : 0.f; x.z = x.z > 0.f ? x.z : 0.f; x.w = x.w > 0.f ? x.w : 0.f;
            } else if (act == 2) {
              x.x = x.x > 0.f ? x.x : x.x * O.slope; x.y = x.y > 0.f ? x.y : x.y * O.slope;
              x.z = x.z > 0.f ? x.z : x.z * O.slope; x.w = x.w > 0.f ? x.w : x.w * O.slope;
            }
            if (has_res && O.mask_last) {  // (a load and its use inside ONE uniform branch: launches without a residual carry no wait)
              const float4 r4 = m2d_bload4(M2D_RS(O.residual), voff);
              x.x += r4.x; x.y += r4.y; x.z += r4.z; x.w += r4.w;
            }
            const unsigned kb = keep >> (4 * qq);
            x.x *= (kb & 1u) ? 1.f : ms; x.y *= (kb & 2u) ? 1.f : ms; x.z *= (kb & 4u) ? 1.f : ms; x.w *= (kb & 8u) ? 1.f : ms;
            if (has_res && !O.mask_last) {
              const float4 r4 = m2d_bload4(M2D_RS(O.residual), voff);
              const float4 sum = make_float4(x.x + r4.x, x.y + r4.y, x.z + r4.z, x.w + r4.w);
              if (two_out) m2d_bstore4(M2D_RS(O.sum_out), voff, sum);
              else x = sum;
            }
            m2d_bstore4(M2D_RS(out_base), voff, x);
            if (stats) {  // this tile's 32 columns of the row: the 8 lanes that share it; one partial per (row, 32-column tile)
              a1 = ok ? (x.x + x.y) + (x.z + x.w) : 0.f;
              a2 = ok ? (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w) : 0.f;
#pragma unroll
              for (int off = 4; off > 0; off >>= 1) {
                a1 += __shfl_xor(a1, off, 64);
                a2 += __shfl_xor(a2, off, 64);
              }
              if ((lane & 7) == 0 && row < p.M) {
                float* dst = O.row_part + ((size_t)(((n0 / BN) * WN + wn) * TN + j0 + jj) * p.M + row) * 2;
                dst[0] = a1;
                dst[1] = a2;
              }
            }
          } else {
            float x = wl[jj * 1024 + rl * 32 + l31];
            if (redirect && (NT > 1 && jj ? red_t[NT - 1] : red_t[0]) && row < p.M) O.col_out[row] = x;  // raw: a bias gradient
            if (act == 1) x = x > 0.f ? x : 0.f;
            else if (act == 2) x = x > 0.f ? x : x * O.slope;
            if (has_res && O.mask_last) x += m2d_bload(M2D_RS(O.residual), voff, 0);
            x *= ((keep >> qq) & 1u) ? 1.f : ms;
            if (has_res && !O.mask_last) {
              const float sum = x + m2d_bload(M2D_RS(O.residual), voff, 0);
              if (two_out) m2d_bstore1(M2D_RS(O.sum_out), voff, sum);
              else x = sum;
            }
            m2d_bstore1(M2D_RS(out_base), voff, x);
            if (stats) {  // the 32 lanes (columns) that share this row; one partial per (row, 32-column tile)
              a1 = ok ? x : 0.f;
              a2 = ok ? x * x : 0.f;
#pragma unroll
              for (int off = 16; off > 0; off >>= 1) {
                a1 += __shfl_xor(a1, off, 64);
                a2 += __shfl_xor(a2, off, 64);
              }
              if (l31 == 0 && row < p.M) {
                float* dst = O.row_part + ((size_t)(((n0 / BN) * WN + wn) * TN + j0 + jj) * p.M + row) * 2;
                dst[0] = a1;
                dst[1] = a2;
              }
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();  // the image is read before the next tiles overwrite it (one wave, in order)
    }
  }
#undef M2D_RS
}

// Split-K in ONE launch (p.tickets != NULL). Every workgroup stores its partial tile as a register image - the slab
// is [tile][split][16 TM TN / 4 float4][256 threads], 16-byte write-through stores - drains, and takes a ticket for its
// tile; the workgroup that draws the last one reads all `splits` images back (device-scope loads, split order, starting
// from zero: the sum does not depend on who arrived last), zeroes the ticket for the next launch and returns true: its
// accumulators then hold the whole K and the ordinary epilogue runs. Everybody else returns false and leaves.
// (The separate m2d_splitk_reduce_kernel costs a launch gap plus 8-35 us and reads every slab from a cold grid; here
// the partials are read by ONE workgroup per tile while they are still in the cache hierarchy.) Launcher: splits <= 16 (more: the separate reduction kernel, one workgroup reading that many images is the slower way).
template <int BM, int BN, int NW = 4>
__device__ __forceinline__ bool m2d_splitk_fixup(const M2dGemmParams& p, int split, int tid, volatile int* flag,
                                                 f32x16 (&acc)[M2dTiling<BM, BN, NW>::TM][M2dTiling<BM, BN, NW>::TN]) {
  constexpr int TM = M2dTiling<BM, BN, NW>::TM;
  constexpr int TN = M2dTiling<BM, BN, NW>::TN;
  constexpr unsigned TILE_BYTES = BM * BN * 4;
  constexpr int PLANE = M2dTiling<BM, BN, NW>::NTH * 16;   // bytes of one float4 per thread: the image's plane pitch
  // (`flag`: a word of the stage buffers, free after the last chunk - a __shared__ of its own would be the 4 bytes that
  // push the LDS-direct kernels from five to four workgroups per CU)
  const unsigned tile = blockIdx.y * gridDim.x + blockIdx.x;
  const unsigned slab_bytes = gridDim.x * gridDim.y * (unsigned)p.splits * TILE_BYTES;  // < 2^31: launcher-checked
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.slab, (short)0, (int)slab_bytes, 0x00020000);
  const unsigned mine = (tile * (unsigned)p.splits + (unsigned)split) * TILE_BYTES + (unsigned)tid * 16u;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        m2d_vf32x4 v;
        v[0] = acc[i][j][4 * r4]; v[1] = acc[i][j][4 * r4 + 1]; v[2] = acc[i][j][4 * r4 + 2]; v[3] = acc[i][j][4 * r4 + 3];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(m2d_u32x4, v), rs,
                                               (int)(mine + (unsigned)(((i * TN + j) * 4 + r4) * PLANE)), 0, 16 /* sc1 */);
      }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned n = __hip_atomic_fetch_add(p.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = (n + 1u == (unsigned)p.splits) ? 1 : 0;
  }
  __syncthreads();
  const bool last = *flag != 0;
  __syncthreads();  // (the epilogue reuses the stage buffers)
  if (!last) return false;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      for (int s = 0; s < p.splits; ++s) {
        const unsigned from = (tile * (unsigned)p.splits + (unsigned)s) * TILE_BYTES + (unsigned)tid * 16u;
        m2d_vf32x4 f[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
          // (whole-vector bit cast: an element-wise cast of this builtin's result is narrowed to one dword load)
          f[r4] = __builtin_bit_cast(m2d_vf32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      rs, (int)(from + (unsigned)(((i * TN + j) * 4 + r4) * PLANE)), 0, 16 /* sc1 */));
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          acc[i][j][4 * r4] += f[r4][0]; acc[i][j][4 * r4 + 1] += f[r4][1];
          acc[i][j][4 * r4 + 2] += f[r4][2]; acc[i][j][4 * r4 + 3] += f[r4][3];
        }
      }
    }
  if (tid == 0) p.tickets[tile] = 0u;  // (visible to the next launch of the stream at the kernel boundary)
  return true;
}

#ifdef M2D_STAMP  // diagnostic builds only (tools/phase_stamps.py): where does a workgroup of the LDS-direct kernel spend its time?
// per workgroup: entries 0..3 = s_memrealtime (100 MHz) at entry / loop entry / loop exit / end, 4..7 = s_memtime (the
// shader clock's counter) at the same points: clock in GHz = d memtime / d realtime / 10 (round 6)
__device__ unsigned long long m2d_stamp_buf[8192 * 8];
#define M2D_STAMP_AT(i)                                                                                      \
  do {                                                                                                       \
    if (threadIdx.x == 0) {                                                                                  \
      unsigned long long* sb__ = m2d_stamp_buf + ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 8192 * 8; \
      sb__[(i)] = __builtin_amdgcn_s_memrealtime();                                                          \
      sb__[4 + (i)] = __builtin_amdgcn_s_memtime();                                                          \
    }                                                                                                        \
  } while (0)
extern "C" int m2d_debug_stamps_reset(void) {
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(m2d_stamp_buf)) != hipSuccess) return -2;
  return hipMemset(p, 0, sizeof(unsigned long long) * 8192 * 8) == hipSuccess ? 0 : -2;
}
extern "C" int m2d_debug_stamps(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(m2d_stamp_buf), (size_t)n * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#else
#define M2D_STAMP_AT(i) do { } while (0)
#endif

template <int BM, int BN, bool AKF, bool BKF, bool MASKED, bool WIDE>
__global__ void __launch_bounds__(256, 2) m2d_gemm_kernel(const M2dGemmParams p) {
  constexpr int LDA = BM + M2D_LDPAD;
  constexpr int LDB = BN + M2D_LDPAD;
  constexpr int WM = BM >= 64 ? 2 : 1;
  constexpr int WN = 4 / WM;
  constexpr int TM = BM / (32 * WM);
  constexpr int TN = BN / (32 * WN);
  constexpr int STAGE = M2D_BK * (LDA + LDB);
  __shared__ float smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM;
  const int wn = wave / WM;
  const int l31 = lane & 31;
  const int lh = lane >> 5;
  M2D_STAMP_AT(0);

  M2dOperand A = p.A;
  M2dOperand B = p.B;
  M2dOutMap O = p.O;
  int N = p.N;
  int nhi = p.nhi;
  int split = blockIdx.z;
  int bx, by;
  m2d_tile_of(p.tile_map, bx, by);
  if (p.bwd_data) {
    // conv backward-data: output phase r of the stride-s lattice uses taps r, r+s, ...
    //   dx[n, ci, s*q + r - pad] = sum_{t, co} Wp[ci, r + s*t, co] * dy[n, co, q - t]
    // (K = (t, co): hi = t, lo = co; the host filled the strides for that order)
    const int r = blockIdx.z;
    const int s = p.phases;
    split = 0;
    const int taps = r < p.ph_ks ? (p.ph_ks - r + s - 1) / s : 0;
    const int qmin = r >= p.ph_pad ? 0 : (p.ph_pad - r + s - 1) / s;
    const int top = p.ph_L - 1 + p.ph_pad - r;
    const int nq = top >= 0 ? (top / s - qmin + 1) : 0;
    if (nq <= 0) return;
    N = p.ph_batch * nq;
    if ((int)(bx * BN) >= N) return;
    nhi = taps;
    A.r_off += r * (p.ph_a_step ? p.ph_a_step : p.ph_cout);
    B.nrows = N;
    B.rdiv = nq;
    B.rdiv_inv = 1.f / (float)nq;
    B.r_off = qmin;
    B.r_pos_off = qmin;
    O.cdiv = nq;
    O.cdiv_inv = B.rdiv_inv;
    O.c_off = s * qmin + r - p.ph_pad;
    O.c_pos_off = O.c_off;
  }

  const int m0 = by * BM;
  const int n0 = bx * BN;

  TileMap<AKF, BM, MASKED> ta;
  TileMap<BKF, BN, MASKED> tb;
  ta.prep(A, m0, tid);
  tb.prep(B, n0, tid);
  const __amdgpu_buffer_rsrc_t ra = m2d_rsrc(A.base, A.nbytes);
  const __amdgpu_buffer_rsrc_t rb = m2d_rsrc(B.base, B.nbytes);
  // an operand without a mask reads its "mask" through a 0-record descriptor (always 0.0f)
  // and uses slope 1, so the masked kernel needs no per-element branch
  const __amdgpu_buffer_rsrc_t rma = m2d_rsrc(A.mask ? A.mask : A.base, A.mask ? A.nbytes : 0u);
  const __amdgpu_buffer_rsrc_t rmb = m2d_rsrc(B.mask ? B.mask : B.base, B.mask ? B.nbytes : 0u);
  if (!A.mask) A.mask_slope = 1.f;
  if (!B.mask) B.mask_slope = 1.f;

  f32x16 acc[TM][TN];
  m2d_acc_init<BM, BN>(p, O, N, split, m0, n0, wm, wn, l31, lh, acc);

  const int cph = (p.kdiv + M2D_BK - 1) / M2D_BK;  // chunks per hi
  const int nchunks = nhi * cph;
  const int cps = (nchunks + p.splits - 1) / p.splits;
  const int c0 = split * cps;
  const int c1 = (c0 + cps < nchunks) ? (c0 + cps) : nchunks;

  if (c0 < c1) {
    ChunkCursor cc;
    cc.kdiv = p.kdiv;
    cc.nhi = nhi;
    cc.lo_outer = p.lo_outer;
    cc.seek(c0, cph);
    cc.a_lo = A.k_safe_lo; cc.a_hi = A.k_safe_hi;
    cc.b_lo = B.k_safe_lo; cc.b_hi = B.k_safe_hi;
    // chunk c0 through the general path (once per block)
    ta.template load<false>(A, ra, rma, cc.hi, cc.lo0, cc.extent());
    tb.template load<false>(B, rb, rmb, cc.hi, cc.lo0, cc.extent());
    ta.template store<LDA>(A, smem, tid);
    tb.template store<LDB>(B, smem + M2D_BK * LDA, tid);
    cc.next();
    __syncthreads();
    M2D_STAMP_AT(1);
    for (int c = c0; c < c1; ++c) {
      const int cur = (c - c0) & 1;
      // Stage chunk c + 1 (cursor cc) while chunk c is multiplied: its buffer loads are issued
      // first (a uniform chunk costs one load per element and a handful of address
      // instructions per chunk, so the burst is short), the MFMAs of chunk c run while they are
      // in flight. The chunk after the last one either belongs to the next split (real data) or
      // has hi == nhi and loads zeros through the general path; its LDS store is harmless.
      if (cc.uniform()) {
        ta.template load<true>(A, ra, rma, cc.hi, cc.lo0, cc.kdiv);
        tb.template load<true>(B, rb, rmb, cc.hi, cc.lo0, cc.kdiv);
      } else {
        ta.template load<false>(A, ra, rma, cc.hi, cc.lo0, cc.extent());
        tb.template load<false>(B, rb, rmb, cc.hi, cc.lo0, cc.extent());
      }
      m2d_chunk_mma<BM, BN, AKF, BKF, MASKED>(ta, tb, A, B, smem, cur, tid, wm, wn, l31, lh, acc);
      cc.next();
      __syncthreads();
    }
  }

  // (two instantiations, not a branch: with both epilogues in one kernel every accumulator stays live across the
  // choice and the kernel needs 30 more registers, i.e. one resident wave per SIMD fewer)
  M2D_STAMP_AT(2);
  if (p.tickets && !m2d_splitk_fixup<BM, BN>(p, split, tid, reinterpret_cast<volatile int*>(smem), acc)) return;
  m2d_tile_epilogue<BM, BN, WIDE>(p, O, N, split, m0, n0, wm, wn, lane, smem + wave * (BM == 128 ? 2048 : 1024), acc);
#ifdef M2D_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  M2D_STAMP_AT(3);
}


// ---- LDS-direct staging variant ------------------------------------------------------------------------------------
// Both operands row-fast and unmasked (conv forward / backward-data over the K-major weight image, plain NN GEMMs):
// the chunk's loads write the LDS image themselves (`buffer_load ... lds`), so the tile needs neither the 16 staging
// registers nor the ds_write pass, and with rows lane-consecutive on both sides the image needs no padding: 32 KB of
// LDS and <= 102 VGPRs per workgroup = FIVE workgroups per CU instead of four.
template <int BM, int BN, int NW = 4>
__device__ __forceinline__ void m2d_chunk_mma_dl(const float* stage, int wm, int wn, int l31, int lh,
                                                 f32x16 (&acc)[M2dTiling<BM, BN, NW>::TM][M2dTiling<BM, BN, NW>::TN]) {
  constexpr int TM = M2dTiling<BM, BN, NW>::TM;
  constexpr int TN = M2dTiling<BM, BN, NW>::TN;
  const float* as = stage + wm * (TM * 32) + l31;
  const float* bs = stage + M2D_BK * BM + wn * (TN * 32) + l31;
  float fa[M2D_BK / 2][TM], fb[M2D_BK / 2][TN];
#pragma unroll
  for (int kk = 0; kk < M2D_BK / 2; ++kk) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[kk][i] = as[(2 * kk + lh) * BM + i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[kk][j] = bs[(2 * kk + lh) * BN + j * 32];
  }
#pragma unroll
  for (int kk = 0; kk < M2D_BK / 2; ++kk)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
  __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
  for (int kk = 0; kk < M2D_BK / 2; ++kk) {
    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
    if (kk < M2D_BK / 2 - 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
  }
}

// Quad tile epilogue (O.quad): the four accumulator registers 4 g .. 4 g + 3 of a lane are rows 4 c .. 4 c + 3 of one
// column, i.e. four consecutive elements of the sub-pixel output - one 16-byte store per register group instead of
// four dword stores a stride of 16 bytes apart. Addresses are 4-byte aligned only (the row offset is s q - pad).
typedef float m2d_f32x4u __attribute__((ext_vector_type(4), aligned(4)));

template <int BM, int BN, int NW = 4>
__device__ __forceinline__ void m2d_tile_epilogue_quad(const M2dGemmParams& p, const M2dOutMap& O, int N, int m0, int n0,
                                                       int wm, int wn, int l31, int lh,
                                                       f32x16 (&acc)[M2dTiling<BM, BN, NW>::TM][M2dTiling<BM, BN, NW>::TN]) {
  constexpr int TM = M2dTiling<BM, BN, NW>::TM;
  constexpr int TN = M2dTiling<BM, BN, NW>::TN;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * (TN * 32) + j * 32 + l31;
    if (col >= N) continue;
    int chi, clo;
    m2d_divmod(col, O.cdiv, O.cdiv_inv, chi, clo);
    const int caddr = chi * O.c_hi_stride + clo * O.c_lo_stride + O.c_off;
    const int pos = clo * O.c_pos_mul + O.c_pos_off;
    const bool whole = pos >= 0 && pos + 3 < O.c_lim;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = m0 + wm * (TM * 32) + i * 32 + 8 * g + 4 * lh;
        if (row >= p.M) continue;
        const int addr = (row >> 2) * O.m_stride + caddr;
        // (signed: a quad that starts left of the row - addr < 0 for the first positions of sample 0 - must not wrap)
        const int maddr = (O.mask_wrap && addr >= (int)O.mask_wrap) ? addr - (int)O.mask_wrap : addr;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[i][j][4 * g + e];
          if (O.bias_mode == 1) x += O.bias[row + e];
          if (O.act == 1) x = x > 0.f ? x : 0.f;
          else if (O.act == 2) x = x > 0.f ? x : x * O.slope;
          v[e] = x;
        }
        if (whole) {
          if (O.mask_last) {
            if (O.residual) {
              const m2d_f32x4u rr = *reinterpret_cast<const m2d_f32x4u*>(O.residual + addr);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += rr[e];
            }
            if (O.mask) {
              const m2d_f32x4u mm = *reinterpret_cast<const m2d_f32x4u*>(O.mask + maddr);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= (mm[e] > 0.f ? 1.f : O.mask_slope);
            }
          } else {
            if (O.mask) {
              const m2d_f32x4u mm = *reinterpret_cast<const m2d_f32x4u*>(O.mask + maddr);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= (mm[e] > 0.f ? 1.f : O.mask_slope);
            }
            if (O.residual) {
              const m2d_f32x4u rr = *reinterpret_cast<const m2d_f32x4u*>(O.residual + addr);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += rr[e];
            }
          }
          m2d_f32x4u o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = v[e];
          *reinterpret_cast<m2d_f32x4u*>(O.out + addr) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if ((unsigned)(pos + e) >= (unsigned)O.c_lim) continue;
            float x = v[e];
            // (element by element: a quad that hangs over the left end of a row starts at addr < its sample's origin)
            const int me = (O.mask_wrap && addr + e >= (int)O.mask_wrap) ? addr + e - (int)O.mask_wrap : addr + e;
            if (O.mask_last) {
              if (O.residual) x += O.residual[addr + e];
              if (O.mask) x *= (O.mask[me] > 0.f ? 1.f : O.mask_slope);
            } else {
              if (O.mask) x *= (O.mask[me] > 0.f ? 1.f : O.mask_slope);
              if (O.residual) x += O.residual[addr + e];
            }
            O.out[addr + e] = x;
          }
        }
      }
    }
  }
}

// EPI: 0 dword epilogue, 1 wide (16-byte rows through LDS), 2 quad (sub-pixel rows)
// (eight waves: 8 waves per SIMD asked for - <= 64 VGPRs AND <= 96 SGPRs; at the 106 SGPRs the four-wave kernels use a
// SIMD's 800 scalar registers hold 7 waves, i.e. only THREE eight-wave workgroups per CU: measured as a second dispatch
// round, 548 -> 588 us on the 960-tile encoder conv)
template <int BM, int BN, int EPI, int NW = 4>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 8 : 2) m2d_gemm_dl_kernel(const M2dGemmParams p) {
  constexpr int WM = M2dTiling<BM, BN, NW>::WM;
  constexpr int TM = M2dTiling<BM, BN, NW>::TM;
  constexpr int TN = M2dTiling<BM, BN, NW>::TN;
  constexpr int NTH = M2dTiling<BM, BN, NW>::NTH;
  static_assert(NW == 4 || (NW == 8 && BM == 128 && BN == 128), "eight waves: the 128 x 128 tile");
  constexpr int STAGE = M2D_BK * (BM + BN);
  __shared__ float smem[2 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM;
  const int wn = wave / WM;
  const int l31 = lane & 31;
  const int lh = lane >> 5;
  M2D_STAMP_AT(0);

  M2dOperand A = p.A;
  M2dOperand B = p.B;
  M2dOutMap O = p.O;
  int N = p.N;
  int nhi = p.nhi;
  int split = blockIdx.z;
  int bx, by;
  m2d_tile_of(p.tile_map, bx, by);
  if (p.bwd_data) {  // as in m2d_gemm_kernel
    const int r = blockIdx.z;
    const int s = p.phases;
    split = 0;
    const int taps = r < p.ph_ks ? (p.ph_ks - r + s - 1) / s : 0;
    const int qmin = r >= p.ph_pad ? 0 : (p.ph_pad - r + s - 1) / s;
    const int top = p.ph_L - 1 + p.ph_pad - r;
    const int nq = top >= 0 ? (top / s - qmin + 1) : 0;
    if (nq <= 0) return;
    N = p.ph_batch * nq;
    if ((int)(bx * BN) >= N) return;
    nhi = taps;
    A.r_off += r * (p.ph_a_step ? p.ph_a_step : p.ph_cout);
    B.nrows = N;
    B.rdiv = nq;
    B.rdiv_inv = 1.f / (float)nq;
    B.r_off = qmin;
    B.r_pos_off = qmin;
    O.cdiv = nq;
    O.cdiv_inv = B.rdiv_inv;
    O.c_off = s * qmin + r - p.ph_pad;
    O.c_pos_off = O.c_off;
  }
  const int m0 = by * BM;
  const int n0 = bx * BN;
  TileMap<false, BM, false, NTH> ta;
  TileMap<false, BN, false, NTH> tb;
  ta.prep_dl(A, m0, tid);
  tb.prep_dl(B, n0, tid);
  const __amdgpu_buffer_rsrc_t ra = m2d_rsrc(A.base, A.nbytes);
  const __amdgpu_buffer_rsrc_t rb = m2d_rsrc(B.base, B.nbytes);

  f32x16 acc[TM][TN];
  if constexpr (EPI == 2) {  // (the quad epilogue adds its bias itself)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  } else {
    m2d_acc_init<BM, BN, NW>(p, O, N, split, m0, n0, wm, wn, l31, lh, acc);
  }

  const int cph = (p.kdiv + M2D_BK - 1) / M2D_BK;
  const int nchunks = nhi * cph;
  const int cps = (nchunks + p.splits - 1) / p.splits;
  const int c0 = split * cps;
  const int c1 = (c0 + cps < nchunks) ? (c0 + cps) : nchunks;

  if (c0 < c1) {
    ChunkCursor cc;
    cc.kdiv = p.kdiv;
    cc.nhi = nhi;
    cc.lo_outer = p.lo_outer;
    cc.seek(c0, cph);
    cc.a_lo = A.k_safe_lo; cc.a_hi = A.k_safe_hi;
    cc.b_lo = B.k_safe_lo; cc.b_hi = B.k_safe_hi;
    ta.template load_lds<false, BM>(A, ra, cc.hi, cc.lo0, cc.extent(), smem, tid);
    tb.template load_lds<false, BN>(B, rb, cc.hi, cc.lo0, cc.extent(), smem + M2D_BK * BM, tid);
    cc.next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    M2D_STAMP_AT(1);
    for (int c = c0; c < c1; ++c) {
      const int cur = (c - c0) & 1;
      // chunk c + 1 streams into the other stage (every wave left it at the barrier that ended chunk c - 1) while
      // chunk c is multiplied; the chunk after the last one loads zeros or the next split's data, harmlessly
      float* nxt = smem + (cur ^ 1) * STAGE;
#ifdef M2D_X_UNIFORM_ONLY  // experiment: the loop without its general path (valid for plain GEMMs with K % 16 == 0)
      {
        ta.template load_lds<true, BM>(A, ra, cc.hi, cc.lo0, cc.kdiv, nxt, tid);
        tb.template load_lds<true, BN>(B, rb, cc.hi, cc.lo0, cc.kdiv, nxt + M2D_BK * BM, tid);
      }
#else
      if (cc.uniform()) {
        ta.template load_lds<true, BM>(A, ra, cc.hi, cc.lo0, cc.kdiv, nxt, tid);
        tb.template load_lds<true, BN>(B, rb, cc.hi, cc.lo0, cc.kdiv, nxt + M2D_BK * BM, tid);
      } else {
        ta.template load_lds<false, BM>(A, ra, cc.hi, cc.lo0, cc.extent(), nxt, tid);
        tb.template load_lds<false, BN>(B, rb, cc.hi, cc.lo0, cc.extent(), nxt + M2D_BK * BM, tid);
      }
#endif
      m2d_chunk_mma_dl<BM, BN, NW>(smem + cur * STAGE, wm, wn, l31, lh, acc);
#ifdef M2D_X_SIMPLE_CURSOR  // experiment: plain GEMM cursor (nhi = 1)
      cc.lo0 += M2D_BK;
#else
      cc.next();
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  // (two instantiations, not a branch: with both epilogues in one kernel every accumulator stays live across the
  // choice and the kernel needs 30 more registers, i.e. one resident wave per SIMD fewer)
  M2D_STAMP_AT(2);
  if (p.tickets && !m2d_splitk_fixup<BM, BN, NW>(p, split, tid, reinterpret_cast<volatile int*>(smem), acc)) return;
  if constexpr (EPI == 2) m2d_tile_epilogue_quad<BM, BN, NW>(p, O, N, m0, n0, wm, wn, l31, lh, acc);
  else m2d_tile_epilogue<BM, BN, EPI == 1, NW>(p, O, N, split, m0, n0, wm, wn, lane, smem + wave * ((BM == 128 && NW == 4) ? 2048 : 1024), acc);
#ifdef M2D_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  M2D_STAMP_AT(3);
}

// ---- sub-pixel backward-data without phantom taps (O.quad == 2, p.tall_last_rb) -------------------------------------
// The sub-pixel form of a stride-4 k25 backward-data has 7 tap slots per phase and 25 taps: slot 6 exists for phase 0
// only, so with rows (ci, r) 3 of every 4 rows of the slot-6 chunks multiply zeros - a seventh of K at a quarter of
// the work, 10.7 % of the launch's MFMAs (DESIGN.md 3.1d "what is left" (2)). Here the 128 rows of the tile are ordered
// phase-major, row = 32 r + ci: the four 32-row MFMA blocks ARE the four phases, and a slot-6 chunk issues the MFMAs of
// block 0 only. For that every wave owns all four row blocks of 32 columns (TM = 4, TN = 1 instead of 2 x 2: five
// fragment reads per four MFMAs instead of four) - and then register i of the four accumulators of a lane is one
// (ci, position) at r = 0..3: the 16-byte store of the quad epilogue comes straight out of the registers.
// K order (lo_outer): co block outer, tap slot inner - a slot-6 chunk every nhi chunks; two inner loops (a wave-uniform
// branch around the MFMAs would keep every accumulator live across it: 700 spilled registers in the k4 kernel's first form).
template <int RB>
__device__ __forceinline__ void m2d_chunk_mma_tall(const float* stage, int wn, int l31, int lh, f32x16 (&acc)[4]) {
  const float* as = stage + l31;
  const float* bs = stage + M2D_BK * 128 + wn * 32 + l31;
  float fa[M2D_BK / 2][RB], fb[M2D_BK / 2];
#pragma unroll
  for (int kk = 0; kk < M2D_BK / 2; ++kk) {
#pragma unroll
    for (int i = 0; i < RB; ++i) fa[kk][i] = as[(2 * kk + lh) * 128 + i * 32];
    fb[kk] = bs[(2 * kk + lh) * 128];
  }
#pragma unroll
  for (int kk = 0; kk < M2D_BK / 2; ++kk)
#pragma unroll
    for (int i = 0; i < RB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i], fb[kk], acc[i], 0, 0, 0);
  __builtin_amdgcn_sched_group_barrier(0x100, RB + 1, 0);
#pragma unroll
  for (int kk = 0; kk < M2D_BK / 2; ++kk) {
    __builtin_amdgcn_sched_group_barrier(0x008, RB, 0);
    if (kk < M2D_BK / 2 - 2) __builtin_amdgcn_sched_group_barrier(0x100, RB > 1 ? 3 : 2, 0);
  }
}

__global__ void __launch_bounds__(256, 2) m2d_gemm_dl_tall_kernel(const M2dGemmParams p) {
  constexpr int BM = 128, BN = 128;
  constexpr int STAGE = M2D_BK * (BM + BN);
  __shared__ float smem[2 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = tid >> 6;
  const int l31 = lane & 31;
  const int lh = lane >> 5;
  M2D_STAMP_AT(0);
  const M2dOperand& A = p.A;
  const M2dOperand& B = p.B;
  const M2dOutMap& O = p.O;
  const int N = p.N;
  int bx, by;
  m2d_tile_of(p.tile_map, bx, by);
  const int m0 = by * BM;
  const int n0 = bx * BN;
  TileMap<false, BM, false, 256> ta;
  TileMap<false, BN, false, 256> tb;
  ta.prep_dl(A, m0, tid);
  tb.prep_dl(B, n0, tid);
  const __amdgpu_buffer_rsrc_t ra = m2d_rsrc(A.base, A.nbytes);
  const __amdgpu_buffer_rsrc_t rb = m2d_rsrc(B.base, B.nbytes);
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int nhi = p.nhi;
  const int cph = (p.kdiv + M2D_BK - 1) / M2D_BK;   // co blocks
  ChunkCursor cc;
  cc.kdiv = p.kdiv;
  cc.nhi = nhi;
  cc.lo_outer = 1;
  cc.seek(0, cph);
  cc.a_lo = A.k_safe_lo; cc.a_hi = A.k_safe_hi;
  cc.b_lo = B.k_safe_lo; cc.b_hi = B.k_safe_hi;
  ta.template load_lds<false, BM>(A, ra, cc.hi, cc.lo0, cc.extent(), smem, tid);
  tb.template load_lds<false, BN>(B, rb, cc.hi, cc.lo0, cc.extent(), smem + M2D_BK * BM, tid);
  cc.next();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  M2D_STAMP_AT(1);
  int cur = 0;
  // the chunk after the one being multiplied streams into the other stage (the chunk after the last one loads zeros)
  auto stage_next = [&]() {
    float* nxt = smem + (cur ^ 1) * STAGE;
    if (cc.uniform()) {
      ta.template load_lds<true, BM>(A, ra, cc.hi, cc.lo0, cc.kdiv, nxt, tid);
      tb.template load_lds<true, BN>(B, rb, cc.hi, cc.lo0, cc.kdiv, nxt + M2D_BK * BM, tid);
    } else {
      ta.template load_lds<false, BM>(A, ra, cc.hi, cc.lo0, cc.extent(), nxt, tid);
      tb.template load_lds<false, BN>(B, rb, cc.hi, cc.lo0, cc.extent(), nxt + M2D_BK * BM, tid);
    }
  };
  auto step_done = [&]() {
    cc.next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  };
  for (int blk = 0; blk < cph; ++blk) {
    for (int t = 0; t < nhi - 1; ++t) {   // slots every phase has
      stage_next();
      m2d_chunk_mma_tall<4>(smem + cur * STAGE, wn, l31, lh, acc);
      step_done();
    }
    stage_next();                          // the last slot: phase 0 only
    m2d_chunk_mma_tall<1>(smem + cur * STAGE, wn, l31, lh, acc);
    step_done();
  }
  M2D_STAMP_AT(2);
  // epilogue: register g of the four accumulators = (ci, position) at r = 0..3
  {
    const int col = n0 + wn * 32 + l31;
    if (col < N) {
      int chi, clo;
      m2d_divmod(col, O.cdiv, O.cdiv_inv, chi, clo);
      const int caddr = chi * O.c_hi_stride + clo * O.c_lo_stride + O.c_off;
      const int pos = clo * O.c_pos_mul + O.c_pos_off;
      const bool whole = pos >= 0 && pos + 3 < O.c_lim;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int ci = (m0 >> 2) + (g & 3) + 8 * (g >> 2) + 4 * lh;
        const int addr = ci * O.m_stride + caddr;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[e][g];
        if (whole) {
          // (signed: a quad that starts left of the row - addr < 0 for the first positions of sample 0 - must not wrap)
          const int maddr = (O.mask_wrap && addr >= (int)O.mask_wrap) ? addr - (int)O.mask_wrap : addr;
          if (O.residual) {
            const m2d_f32x4u rr = *reinterpret_cast<const m2d_f32x4u*>(O.residual + addr);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rr[e];
          }
          if (O.mask) {
            const m2d_f32x4u mm = *reinterpret_cast<const m2d_f32x4u*>(O.mask + maddr);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= (mm[e] > 0.f ? 1.f : O.mask_slope);
          }
          m2d_f32x4u o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = v[e];
          *reinterpret_cast<m2d_f32x4u*>(O.out + addr) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if ((unsigned)(pos + e) >= (unsigned)O.c_lim) continue;
            float x = v[e];
            const int me = (O.mask_wrap && addr + e >= (int)O.mask_wrap) ? addr + e - (int)O.mask_wrap : addr + e;
            if (O.residual) x += O.residual[addr + e];
            if (O.mask) x *= (O.mask[me] > 0.f ? 1.f : O.mask_slope);
            O.out[addr + e] = x;
          }
        }
      }
    }
  }
#ifdef M2D_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  M2D_STAMP_AT(3);
}

// LDS byte address of a pointer into a __shared__ array, and a 16-byte LDS read the compiler does not see as a memory
// access (see m2d_conv_k4_kernel)
typedef float m2d_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned m2d_lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
}
__device__ __forceinline__ m2d_f32x4 m2d_ds_read_b128(unsigned addr) {
  m2d_f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

// One 16-deep chunk of the tap-vectorised kernel out of LDS: slots [LO, HI) of its four tap groups (all four = [0, 4)).
// Fragment reads through inline asm: hipcc otherwise puts an `s_waitcnt vmcnt(0)` in front of the first ds_read of the
// chunk (it cannot tell that the LDS-DMA just issued fills the OTHER stage), which serialises staging and multiplying.
// The ordering the hardware needs is explicit: this stage was filled, waited for (vmcnt(0)) and fenced by the barrier
// at the end of the previous iteration; its reads are waited for below before the MFMAs.
template <int BM, int BN, int LO, int HI>
__device__ __forceinline__ void m2d_k4_mma(const float* stage, int wm, int wn, int l31, int lh,
                                           f32x16 (&acc)[BM / 64][BN / 64]) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const unsigned as = m2d_lds_addr(stage + (wm * (TM * 32) + l31) * 4);
  const unsigned bs = m2d_lds_addr(stage + 16 * BM + (wn * (TN * 32) + l31) * 4);
  // 64-row tiles read both 8-deep halves of the chunk up front (24 registers); 128-row tiles one half at a time:
  // 32 fragment registers beside 64 accumulators would not fit the 96 registers of five waves per SIMD
  constexpr bool UPFRONT = BM == 64;
  m2d_f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    if (UPFRONT ? s2 == 0 : true) {
#pragma unroll
      for (int h2 = (UPFRONT ? 0 : s2); h2 < (UPFRONT ? 2 : s2 + 1); ++h2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[UPFRONT ? h2 : 0][i] = m2d_ds_read_b128(as + (((2 * h2 + lh) * BM + i * 32) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[UPFRONT ? h2 : 0][j] = m2d_ds_read_b128(bs + (((2 * h2 + lh) * BN + j * 32) << 4));
      }
    }
    // reads return in order: the first half's TM + TN are in when at most the second half's are outstanding
    if (UPFRONT && s2 == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TM + TN) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // every register of a fragment quad stays allocated until its read has landed: the compiler takes the asm read's
    // result as present at once, and with slots left out (LO > 0 or HI < 4) it would hand the unused registers of
    // the quad to something else while the LDS data is still on its way into them
    if constexpr (LO > 0 || HI < 4) {
#pragma unroll
      for (int h2 = 0; h2 < (UPFRONT ? 2 : 1); ++h2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[h2][i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(fb[h2][j]));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int m = LO; m < HI; ++m)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[UPFRONT ? s2 : 0][i][m], fb[UPFRONT ? s2 : 0][j][m], acc[i][j], 0, 0, 0);
    if (!UPFRONT) __builtin_amdgcn_sched_barrier(0);  // the second half's reads reuse the fragment registers
  }
}

// ---- tap-vectorised stride-4 forward conv ----------------------------------------------------------------------------
// y[n, co, l] = sum_{ci, t} W[co, ci, t] x[n, ci, 4 l + t - pad]  (the audio critic's k25 / s4 layers,
// phase3/archis/default.py:298-303, and the forward-mode tangent of the penalty through them).
// With stride 4 the four taps t' = 4 g .. 4 g + 3 (t' = t + P - pad, P = pad rounded up to a multiple of 4) of one
// output position are 16 contiguous, 16-byte ALIGNED bytes of x, and consecutive output positions are exactly 16
// bytes apart: ONE `buffer_load_dwordx4 ... lds` per wave moves 64 rows x 4 taps as a contiguous kilobyte - where the
// generic engine gathers the same floats as four dword loads that each touch a quarter of eight cache lines.
// The LDS image is [tap group][row][4 taps]; a lane's fragment read is one ds_read_b128 (4 k's) instead of four
// ds_read_b32, and the MFMA k-pairing follows (lanes 0-31 supply k = 8 s + m, lanes 32-63 k = 8 s + 4 + m: any
// pairing is a valid contraction order as long as both operands use it). Per wave and 16-deep chunk: 4 LDS-DMA
// instructions and 8 fragment reads for 32 MFMAs (generic LDS-direct kernel: 16 and 32).
// K order: group index kg = ci * NG + g, NG = ceil((ks + P - pad) / 4) groups per channel (k25, pad 11: 7 groups = 28
// taps, the 3 phantom taps carry zero weights: 12 % more MFMA work than the 25 real taps, paid back by the loads);
// weights come pre-packed as Wk4[kg][Cout][4] (m2d_conv1d_pack_weights_k4). Groups never straddle the row ends
// because L % 4 == 0 (launcher-checked), so padding is a per-lane out-of-range offset, as elsewhere.
template <int BM, int BN, bool WIDE>
__global__ void __launch_bounds__(256, 5) m2d_conv_k4_kernel(const M2dGemmParams p) {
  static_assert(BM == 64 || BM == 128, "k4 kernel: 64- or 128-row tiles");
  constexpr int WM = 2, WN = 2;
  constexpr int TM = BM / (32 * WM);
  constexpr int TN = BN / (32 * WN);
  constexpr int HA = BM / 64, HB = BN / 64;   // 64-row DMA pieces per tap group
  constexpr int STAGE = 16 * (BM + BN);       // floats: 4 groups x (BM + BN) rows x 4 taps
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  // (the wave index as a SCALAR: the tap-group cursor below is wave-uniform state; derived from a vector register it
  // would live in VGPRs and every LDS-DMA would be wrapped in a waterfall loop to make its scalar offset uniform)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int l31 = lane & 31, lh = lane >> 5;
  M2D_STAMP_AT(0);
  const M2dOperand& A = p.A;
  const M2dOperand& B = p.B;
  const int N = p.N;
  int bx, by;
  m2d_tile_of(p.tile_map, bx, by);
  const int m0 = by * BM, n0 = bx * BN;
  const int split = blockIdx.z;
  const __amdgpu_buffer_rsrc_t ra = m2d_rsrc(A.base, A.nbytes);
  const __amdgpu_buffer_rsrc_t rb = m2d_rsrc(B.base, B.nbytes);
  // per-lane constants: wave w stages tap group (4 c + w) of every chunk c, all its 64-row pieces
  unsigned va[HA], vb[HB];
  int posb[HB];
#pragma unroll
  for (int h = 0; h < HA; ++h) {
    const int row = m0 + h * 64 + lane;
    va[h] = row < p.M ? (unsigned)row << 4 : M2D_OOB;
  }
#pragma unroll
  for (int h = 0; h < HB; ++h) {
    const int g = n0 + h * 64 + lane;
    const bool rv = g < N;
    int hi, lo;
    m2d_divmod(rv ? g : 0, B.rdiv, B.rdiv_inv, hi, lo);
    vb[h] = (unsigned)(hi * B.r_hi_stride + lo * B.r_lo_stride + B.r_off) << 2;
    posb[h] = rv ? lo * B.r_pos_mul + B.r_pos_off : M2D_BAD;
  }
  f32x16 acc[TM][TN];
  m2d_acc_init<BM, BN>(p, p.O, N, split, m0, n0, wm, wn, l31, lh, acc);

  const int ng = p.k4_ng;
  const int nkg = p.nhi * ng;                      // nhi = Cin
  const int nchunks = (nkg + 3) >> 2;
  const int cps = (nchunks + p.splits - 1) / p.splits;
  const int c0 = split * cps;
  const int c1 = (c0 + cps < nchunks) ? (c0 + cps) : nchunks;
  // Phantom-paired K order (p.k4_pair; the same order in m2d_pack_weights_k4_kernel): the aligned tap groups of a k25 /
  // pad 11 layer are g0 = (phantom, 0, 1, 2), g1..g5 full, g6 = (23, 24, phantom, phantom) - 3 of 28 slots multiply
  // zeros. The MFMA k-pairing puts slot m of one group beside slot m of the NEXT group of its chunk, so when a chunk holds
  // the g0 groups of four channels (or their g6 groups) whole MFMAs are zero x anything and are left out. K is walked
  // in three regions, each with its own loop: chunks [0, R1) the full groups (channel-major), [R1, R2) the g0 groups
  // (slots 1..3 multiplied), [R2, nchunks) the last groups (slots 0..1): the real taps only. The staging cursor is one
  // function of the chunk index, so the software pipeline runs across the region boundaries.
  const bool pair = p.k4_pair != 0;
  const int nfull = ng - 2;
  const int R1 = pair ? (p.nhi * nfull) >> 2 : nchunks;
  const int R2 = pair ? R1 + (p.nhi >> 2) : nchunks;
  if (c0 < c1) {
    // wave-uniform cursor of THIS wave's tap group of the chunk being staged (chunk index cs): (ci, g), advanced by
    // (dci, dg) per chunk - (0, 4) with wraps where groups are walked channel-major, (4, 0) in the paired order's
    // partial-group regions (one channel per wave and chunk, the group fixed); re-seeded where a region begins
    int cs = c0;
    int kg = 4 * c0 + wave;
    int ci, g, dci = 0, dg = 4;
    const int period = pair ? nfull : ng;
    int glim = pair ? 1 + nfull : ng;   // first group index past the walked range (paired region 1: groups 1..ng-2)
    if (!pair) {
      ci = kg / ng;
      g = kg - ci * ng;
    } else if (c0 < R1) {
      ci = kg / nfull;
      g = 1 + kg - ci * nfull;
    } else {
      ci = 4 * (c0 - (c0 < R2 ? R1 : R2)) + wave;
      g = c0 < R2 ? 0 : ng - 1;
      dci = 4;
      dg = 0;
      glim = 0x7fffffff;
    }
    auto stage = [&](float* st) {
      const bool kok = kg < nkg;
      const int sa = kg * (A.r_lo_stride << 2);                // A.r_lo_stride = Cout * 4 floats per group
      const int sb = (ci * B.k_hi_stride + 4 * g) << 2;
      float* da = st + (wave * BM) * 4;
      float* db = st + 16 * BM + (wave * BN) * 4;
#pragma unroll
      for (int h = 0; h < HA; ++h)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (m2d_lds_f*)(da + h * 256), 16, (int)(kok ? va[h] : M2D_OOB), sa, 0, 0);
#pragma unroll
      for (int h = 0; h < HB; ++h) {
        // (the group's offset is added on the vector side: the hardware range check sees the vector offset only, and
        // a row's own offset is negative for the first positions of the first sample - "x - P" in front of the tensor)
        const bool ok = kok && (unsigned)(posb[h] + 4 * g) < (unsigned)B.lim;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (m2d_lds_f*)(db + h * 256), 16, (int)(ok ? vb[h] + (unsigned)sb : M2D_OOB), 0, 0, 0);
      }
      kg += 4;
      cs += 1;
      if (cs == R1 || cs == R2) {   // (paired order only: R1 = R2 = nchunks otherwise, where nothing real is staged)
        ci = wave;
        g = cs == R1 ? 0 : ng - 1;
        dci = 4;
        dg = 0;
        glim = 0x7fffffff;   // (no wraps: the group is fixed from here on)
      } else {
        ci += dci;
        g += dg;
        if (g >= glim) { g -= period; ci += 1; }
        if (g >= glim) { g -= period; ci += 1; }  // fewer than 4 groups per period: at most two wraps for >= 2
      }
    };
    stage(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    M2D_STAMP_AT(1);
    // one region's chunks [cb, ce): stage chunk c + 1 (past the end: zeros or the next split's data, harmless) while
    // slots [LO, HI) of chunk c are multiplied
#define M2D_K4_REGION(cb, ce, LO, HI)                                                                          \
    for (int c = (cb); c < (ce); ++c) {                                                                        \
      const int cur = (c - c0) & 1;                                                                            \
      stage(smem + (cur ^ 1) * STAGE);                                                                         \
      m2d_k4_mma<BM, BN, LO, HI>(smem + cur * STAGE, wm, wn, l31, lh, acc);                                    \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
      __syncthreads();                                                                                         \
    }
    const int e1 = c1 < R1 ? c1 : R1, b2 = c0 > R1 ? c0 : R1, e2 = c1 < R2 ? c1 : R2, b3 = c0 > R2 ? c0 : R2;
    M2D_K4_REGION(c0, e1, 0, 4)
    M2D_K4_REGION(b2, e2, 1, 4)
    M2D_K4_REGION(b3, c1, 0, 2)
#undef M2D_K4_REGION
  }
  const M2dOutMap& O = p.O;
  M2D_STAMP_AT(2);
  if (p.tickets && !m2d_splitk_fixup<BM, BN>(p, split, tid, reinterpret_cast<volatile int*>(smem), acc)) return;
  m2d_tile_epilogue<BM, BN, WIDE>(p, O, N, split, m0, n0, wm, wn, lane, smem + wave * (BM == 128 ? 2048 : 1024), acc);
#ifdef M2D_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  M2D_STAMP_AT(3);
}

// Sum the split-K slabs in a fixed order (deterministic) and apply the epilogue. Eight
// independent partial chains (slab z goes to chain z % 8, chains combined pairwise) keep eight
// loads per thread in flight: one dependent chain over up to 128 slabs is latency-bound.
__global__ void __launch_bounds__(256) m2d_splitk_reduce_kernel(const M2dGemmParams p) {
  const size_t total = (size_t)p.M * p.N;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int row = (int)(idx / p.N);
    const int col = (int)(idx - (size_t)row * p.N);
    float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* sp = p.slab + idx;
    int z = 0;
    for (; z + 8 <= p.splits; z += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] += sp[(size_t)(z + j) * total];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (z + j < p.splits) c[j] += sp[(size_t)(z + j) * total];
    const float s = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
    if (col + 1 == p.O.redirect_col_p1) {
      p.O.col_out[row] = s;
      continue;
    }
    int chi, clo;
    m2d_divmod(col, p.O.cdiv, p.O.cdiv_inv, chi, clo);
    bool ok = true;
    const int mlo = p.O.m_div > 0 ? row % p.O.m_div : 0;
    if (p.O.c_lim > 0) ok = (unsigned)(clo * p.O.c_pos_mul + p.O.c_pos_off + mlo * p.O.m_pos_mul) < (unsigned)p.O.c_lim;
    if (ok) {
      const int raddr = p.O.m_div > 0 ? (row / p.O.m_div) * p.O.m_stride + mlo * p.O.m_lo_stride : row * p.O.m_stride;
      const int addr = raddr + chi * p.O.c_hi_stride + clo * p.O.c_lo_stride + p.O.c_off;
      p.O.out[addr] = m2d_epilogue(p.O, s, row, col, addr);
    }
  }
}

// sums[2*row .. +1] = sum over p < P of part[(p * M + row) * 2 .. +1], fp64, fixed order: thread t adds
// partials t, t + 256, ... and the block combines the 256 chains pairwise.
__global__ void __launch_bounds__(256) m2d_rowsums_reduce_kernel(const float* __restrict__ part, int P, int M,
                                                                 double* __restrict__ sums) {
  __shared__ double sh[2][256];
  const int row = blockIdx.x, t = threadIdx.x;
  double a = 0.0, b = 0.0;
  for (int q = t; q < P; q += 256) {
    const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)q * M + row) * 2);
    a += (double)v.x;
    b += (double)v.y;
  }
  sh[0][t] = a;
  sh[1][t] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) {
      sh[0][t] += sh[0][t + s];
      sh[1][t] += sh[1][t + s];
    }
    __syncthreads();
  }
  if (t == 0) {
    sums[2 * row] = sh[0][0];
    sums[2 * row + 1] = sh[1][0];
  }
}

// Many partials (the first encoder conv: 15 360 per row; the WaveGAN one: 61 440) make one block per row a
// latency-bound chain of strided 8-byte loads (150 us measured for 61 440 x 32 rows). Two stages instead, both in
// fixed order: stage 1, G blocks, block g owns partials [g * S, (g + 1) * S) for ALL rows - consecutive threads read
// consecutive rows of one partial, i.e. whole lines - and writes one fp64 pair per row; stage 2 adds the G pairs.
__global__ void __launch_bounds__(256) m2d_rowsums_stage1_kernel(const float* __restrict__ part, int P, int M, int S,
                                                                  double* __restrict__ scratch) {
  __shared__ double sh[2][256];
  const int t = threadIdx.x;
  const int p0 = blockIdx.x * S;
  const int p1 = p0 + S < P ? p0 + S : P;
  const int lanes = M < 256 ? M : 256;  // threads along the rows of one partial
  const int npl = 256 / lanes;          // partials in flight side by side (>= 1)
  const int pl = t / lanes, rl = t - pl * lanes;
  for (int r0 = 0; r0 < M; r0 += 256) {
    const int row = r0 + rl;
    double a0 = 0.0, b0 = 0.0, a1 = 0.0, b1 = 0.0;
    if (row < M && pl < npl) {
      int q = p0 + pl;
      for (; q + npl < p1; q += 2 * npl) {
        const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)q * M + row) * 2);
        const float2 w = *reinterpret_cast<const float2*>(part + ((size_t)(q + npl) * M + row) * 2);
        a0 += (double)v.x; b0 += (double)v.y;
        a1 += (double)w.x; b1 += (double)w.y;
      }
      if (q < p1) {
        const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)q * M + row) * 2);
        a0 += (double)v.x; b0 += (double)v.y;
      }
    }
    sh[0][t] = a0 + a1;
    sh[1][t] = b0 + b1;
    __syncthreads();
    if (pl == 0 && row < M) {
      double a = 0.0, b = 0.0;
      for (int j = 0; j < npl; ++j) {
        a += sh[0][j * lanes + rl];
        b += sh[1][j * lanes + rl];
      }
      scratch[((size_t)blockIdx.x * M + row) * 2] = a;
      scratch[((size_t)blockIdx.x * M + row) * 2 + 1] = b;
    }
    __syncthreads();
  }
}

// 8 rows per block, 32 lanes per row: lane j adds groups j, j + 32, ... (independent loads), then the 32 lane sums
// are added in lane order (one thread walking all G groups was a chain of G dependent-latency loads: 76 us for 256)
__global__ void __launch_bounds__(256) m2d_rowsums_stage2_kernel(const double* __restrict__ scratch, int G, int M,
                                                                  double* __restrict__ sums) {
  __shared__ double sh[2][8][33];
  const int rl = threadIdx.x >> 5, j = threadIdx.x & 31;
  const int row = blockIdx.x * 8 + rl;
  double a = 0.0, b = 0.0;
  if (row < M)
    for (int g = j; g < G; g += 32) {
      a += scratch[((size_t)g * M + row) * 2];
      b += scratch[((size_t)g * M + row) * 2 + 1];
    }
  sh[0][rl][j] = a;
  sh[1][rl][j] = b;
  __syncthreads();
  if (j == 0 && row < M) {
    double ta = 0.0, tb = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      ta += sh[0][rl][k];
      tb += sh[1][rl][k];
    }
    sums[2 * row] = ta;
    sums[2 * row + 1] = tb;
  }
}

int m2d_rowsums_reduce(const float* part, int P, int M, double* sums, double* scratch, hipStream_t stream) {
  if (P <= 0 || M <= 0) M2D_FAIL(M2D_ERR_ARG, "m2d_rowsums_reduce: bad arguments");
  if (scratch && (long long)P * M > M2D_ROWSUMS_TWO_STAGE) {
    int G = m2d_ceil_div(P, 64);
    if (G > M2D_ROWSUMS_GROUPS) G = M2D_ROWSUMS_GROUPS;
    const int S = m2d_ceil_div(P, G);
    G = m2d_ceil_div(P, S);
    hipLaunchKernelGGL(m2d_rowsums_stage1_kernel, dim3(G), dim3(256), 0, stream, part, P, M, S, scratch);
    hipLaunchKernelGGL(m2d_rowsums_stage2_kernel, dim3(m2d_ceil_div(M, 8)), dim3(256), 0, stream,
                       (const double*)scratch, G, M, sums);
    M2D_CHECK_LAUNCH("m2d_rowsums_stage_kernels");
    return M2D_OK;
  }
  hipLaunchKernelGGL(m2d_rowsums_reduce_kernel, dim3(M), dim3(256), 0, stream, part, P, M, sums);
  M2D_CHECK_LAUNCH("m2d_rowsums_reduce_kernel");
  return M2D_OK;
}

// Launch plans. A plan is (tile height BM, split-K factor). Candidates are ranked by a small cost
// model (microseconds, fitted to MI355X measurements of the phase-3 layer shapes): 256 CUs, every
// CU works through its share of blocks; a block spends tc(BM) per 16-deep chunk when the CU's
// matrix pipes are kept busy by other resident blocks and more when it is alone (the load -> LDS ->
// MFMA chain of one block is then exposed). Split-K costs one slab round trip through HBM plus a
// second launch. Block counts just above a multiple of 256 leave most CUs idle for the last round,
// which is why the candidate split factors are the ones that land ON a multiple.
// With M2D_AUTOTUNE=1 the launcher TIMES the model's few best candidates the first time it meets a
// shape - on the caller's own operands; every plan writes the same output up to summation order -
// and keeps the fastest. Measured on the phase-3 step (A/B on one box, 3 runs each): 16.21 vs 16.28 ms,
// i.e. the model's first choice is already within 0.5 % of the timed best, so timing is OFF by
// default (plans, and with them summation orders, then do not depend on timing noise).
#define M2D_MAX_CAND 6

struct PlanCand {
  int bm, splits;
  double cost;
};

#define M2D_FUSE_MAX_SPLITS 16
// which cost model ranks the plans (m2d_plan_model_set; M2D_PLAN_MODEL overrides the default once at load): 4 = the
// round-4 model (default: best where a loop body's streams overlap - the two-branch phase-3 critic), 5 = chunk-step floor
// by tile height + splits up to 256 (best where launches run one after the other: phase 2, the pose-only critic)
#include <atomic>
static std::atomic<int> g_plan_model{[] { const char* e = getenv("M2D_PLAN_MODEL"); return (e && e[0] == '5') ? 5 : 4; }()};
extern "C" int m2d_plan_model_set(int model) {
  if (model != 4 && model != 5) M2D_FAIL(M2D_ERR_ARG, "m2d_plan_model_set: 4 or 5");
  g_plan_model.store(model, std::memory_order_relaxed);
  return M2D_OK;
}
extern "C" int m2d_plan_model_get(void) { return g_plan_model.load(std::memory_order_relaxed); }

// Round 5, the rounds model (plan kinds 1 and 2, M2D_PLAN_ROUNDS=0 switches it off): the two launch families whose
// plans the round-4 model got wrong by 8-13 % when swept one by one (tools/plan_sweep.py, 1 900 timed (shape, plan) points
// on MI355X, profiles/r05d_plan_sweep_*.txt) - the K-streaming weight gradients (a handful of output tiles, K = B * Lout in
// the hundreds of thousands) and the strided backward-data launches. What the old formula lacks, in the order it matters:
//  * workgroups run in ROUNDS: a CU holds R = 4 / 7 / 8 of the 128- / 64- / 32-row kernels (LDS, §3.1e), all workgroups of
//    a launch have the same K range, so the first 256 R of them finish together and the rest start on an empty chip:
//    1 200 128-row workgroups are a full round plus 176 lone ones at the lone-workgroup step (the 128 -> 256 backward-data
//    launch: 602 us, as 2 400 64-row workgroups 535 us);
//  * a chunk step of n co-resident workgroups costs a n + c, not max(a n, floor): two 128-row workgroups per CU (what a
//    "512 workgroups fill the chip" plan gives) run at 2.46 us per step = 80 % of the matrix pipe, four at 4.4 us = 89 %;
//  * splits up to 256 (7 tiles x 256 = one full round of the 64-row kernel: the 32 -> 64 weight gradient 695 -> 615 us).
// Parameters: least squares on the sweep (rms 6 % on the weight gradients; every pick within 2 % of the measured best).
struct RoundsModel {
  double a[3], c[3], f[3], P, s0, s1;
};
static const int kRoundsResident[3] = {4, 7, 8};
static const RoundsModel kRoundsBwdW = {{0.859, 0.467, 0.309}, {0.608, 0.343, 0.126}, {1.452, 0.961, 0.748}, 8.3, 5.4, 0.196};
static const RoundsModel kRoundsBwdD = {{0.919, 0.419, 0.301}, {0.404, 0.468, 0.065}, {0.510, 1.307, 0.656}, 12.3, 1.2, 0.126};
static double rounds_cost(const RoundsModel& m, int b, long long W, double cps, long long sp, int M, int N) {
  const int R = kRoundsResident[b];
  auto step = [&](double n) {
    const double v = m.a[b] * n + m.c[b];
    return v > m.f[b] ? v : m.f[b];
  };
  const long long full = W / (256LL * R), rem = W - full * 256LL * R;
  double tot = (double)full * (cps * step((double)R) + m.P);
  if (rem > 0) tot += cps * step((double)m2d_ceil_div64(rem, 256)) + m.P;
  if (sp > 1) tot += m.s0 * (sp > 16 ? 1.0 : 0.3) + m.s1 * (double)sp * (double)M * (double)N * 4.0 / 1.0e6;
  return 6.0 + tot;
}

static int plan_candidates(int M, int N, int nchunks, int phases, bool allow_split, double small_tile_penalty,
                           PlanCand* out, int kind = 0) {
  const int ph = phases > 1 ? phases : 1;
  const bool can_split = allow_split && phases <= 1;
  const long long nt = m2d_ceil_div(N, 128);
  const int bms[3] = {128, 64, 32};
  // us per block-chunk with the matrix pipes busy; `small_tile_penalty` (>= 1) prices operands whose
  // gather is expensive per load (strided conv inputs: 16 B between lanes), which small M tiles re-load more often
  const double pen = small_tile_penalty > 1.0 ? small_tile_penalty : 1.0;
  const double tc[3] = {0.98, 0.56 * pen, 0.33 * pen};
  const int resident[3] = {3, 5, 7};         // blocks per CU (VGPR / LDS budget)
  // Round 5: a chunk step of ONE workgroup cannot be shorter than its staging round trip (issue -> landed -> barrier),
  // whatever the tile height: measured 1.46 us per chunk for a lone 128-row workgroup, 1.2-1.39 us for 64-row ones
  // (two per CU: the audio critic's 32 -> 64 weight gradient ran 73 splits = 2 workgroups per CU at 730 us, 256 splits
  // at 623), 1.2 us for 32-row ones. The old floor, tc + 0.55, priced a 64-row step at 1.11 us and a 32-row one at 0.88,
  // so plans with two or three workgroups per CU looked as good as plans that fill the CU. With the new floor (and splits
  // up to 256) the launches measured one by one get faster - engine 0.583 -> 0.588 of peak, the pose critic 61 -> 64.5
  // TFLOP/s, the 32 -> 64 weight gradient 692 -> 653 us (profiles/r05_planmodel_shapes_diff.txt) - but the STEP, whose
  // streams overlap, gets slower: 12.00 -> 12.10 ms, five alternating pairs (profiles/r05_ab_planmodel.txt): under
  // overlap a launch that leaves CUs idle costs little (another stream's workgroups take them), while extra splits cost
  // slab traffic and fix-up work for everybody. Where nothing overlaps the launches it is the other way round: phase 2
  // 2.199 -> 2.172 ms, the U-Net / pose-only-critic config 14.88 -> 14.72 ms. So the model is a setting
  // (m2d_plan_model_set, once per process before its first launch): the two-branch step keeps 4, bench.py's c2 / c5 presets
  // and the phase-2 train script select 5.
  const bool model5 = g_plan_model.load(std::memory_order_relaxed) == 5;
  static const double slab_mul = [] { const char* e = getenv("M2D_SLAB_COST"); return e ? atof(e) : 1.0; }();   // A/B lever
  const double lchunk[3] = {1.50, 1.35, 1.20};
  static const bool rounds_on = [] { const char* e = getenv("M2D_PLAN_ROUNDS"); return !(e && e[0] == '0'); }();
  const RoundsModel* rm = !rounds_on ? nullptr : kind == M2D_PLAN_BWD_WEIGHT ? &kRoundsBwdW : kind == M2D_PLAN_BWD_DATA ? &kRoundsBwdD : nullptr;
  const long long max_splits = (model5 || rm) ? 256 : 128;
  PlanCand all[40];
  int na = 0;
  for (int b = 0; b < 3; ++b) {
    const long long tiles = (long long)m2d_ceil_div(M, bms[b]) * nt * ph;
    long long cand[10];
    int nc = 0;
    cand[nc++] = 1;
    if (can_split && nchunks >= 8) {
      for (int k = 1; k <= 8; ++k) {
        long long sp = (256LL * k) / tiles;
        if (sp > max_splits) sp = max_splits;
        if (sp > nchunks / 4) sp = nchunks / 4;
        if (sp > 1) cand[nc++] = sp;
      }
    }
    for (int c = 0; c < nc; ++c) {
      const long long sp = cand[c];
      bool dup = false;
      for (int q = 0; q < na; ++q) dup |= all[q].bm == bms[b] && all[q].splits == (int)sp;
      if (dup || na >= 40) continue;
      // a CU holding per_cu blocks, `conc` of them resident at a time: throughput bound
      // per_cu * tc per chunk step, latency bound (one block's load -> LDS -> MFMA chain is
      // ~0.55 us longer than its matrix time) ceil(per_cu / conc) * (tc + 0.55)
      const long long per_cu = m2d_ceil_div64(tiles * sp, 256);
      const long long conc = per_cu < resident[b] ? per_cu : resident[b];
      const double cps = (double)m2d_ceil_div64(nchunks > 0 ? nchunks : 1, sp);
      const double thr = (double)per_cu * tc[b];
      const double step = (model5 && lchunk[b] > tc[b] + 0.55) ? lchunk[b] : tc[b] + 0.55;
      const double lat = (double)m2d_ceil_div64(per_cu, conc) * step;
      double cost = 8.0 + cps * (thr > lat ? thr : lat);
      // (the second launch is gone for splits <= 16 on a stream with tickets, m2d_splitk_fixup, but pricing the split
      // cheaper - 2 or 0 us instead of 6 - moved neither C3 nor C2: the slab term decides)
      if (sp > 1) cost += 6.0 + slab_mul * (double)sp * (double)M * (double)N * 8.0 / 4.0e6;
      if (rm) cost = rounds_cost(*rm, b, tiles * sp, cps, sp, M, N);
      // weight gradients keep the tallest tile that fits M: swept alone, 64-row tiles with twice the splits are 1-3 % faster
      // on the 64 -> 128 and 128 -> 256 layers, but they spend more CU time per flop (1.5x the LDS fill) and the step's other
      // streams pay for it: 11.60 ms without the rounds model, 11.68 with 64-row picks, 11.52-11.56 with this rule
      // (profiles/r05d_ab_rounds.txt)
      if (rm && kind == M2D_PLAN_BWD_WEIGHT && bms[b] < 128 && M >= 128) cost *= 2.0;
      all[na].bm = bms[b];
      all[na].splits = (int)sp;
      all[na].cost = cost;
      ++na;
    }
  }
  for (int i = 0; i < na; ++i)
    for (int j = i + 1; j < na; ++j)
      if (all[j].cost < all[i].cost) {
        PlanCand t = all[i];
        all[i] = all[j];
        all[j] = t;
      }
  int n = 0;
  for (int i = 0; i < na && n < M2D_MAX_CAND; ++i)
    if (all[i].cost <= 1.5 * all[0].cost + 6.0) out[n++] = all[i];
  return n;
}

#ifdef M2D_TUNING
// tuning builds: what the last launch was planned from and with (tools/plan_sweep.py dumps it, tools/plan_fit.py fits
// the rounds model on the dump)
static int g_last_plan[9];
extern "C" int m2d_debug_last_plan(int* out) {
  memcpy(out, g_last_plan, sizeof(g_last_plan));
  return M2D_OK;
}
#endif

// The model's first choice; ws_bytes covers every candidate the launcher may time.
M2dGemmPlan m2d_gemm_plan(int M, int N, int nchunks, int phases, bool allow_split, double small_tile_penalty, int kind) {
  PlanCand c[M2D_MAX_CAND];
  const int n = plan_candidates(M, N, nchunks, phases, allow_split, small_tile_penalty, c, kind);
  M2dGemmPlan pl;
  pl.bm = n ? c[0].bm : 128;
  pl.splits = n ? c[0].splits : 1;
  pl.ws_bytes = 0;
  for (int i = 0; i < n; ++i) {
    const size_t b = m2d_slab_bytes(M, N, c[i].bm, c[i].splits);
    if (b > pl.ws_bytes) pl.ws_bytes = b;
  }
#ifdef M2D_TUNING
  if (const char* e = getenv("M2D_PLAN")) {  // "bm,splits" (tuning builds only)
    int bm = 0, sp = 0;
    if (sscanf(e, "%d,%d", &bm, &sp) == 2) {
      if (bm == 32 || bm == 64 || bm == 128) pl.bm = bm;
      if (sp >= 1 && allow_split && phases <= 1 && sp <= nchunks) pl.splits = sp;
      if (sp == 1) pl.splits = 1;
      const size_t b = m2d_slab_bytes(M, N, pl.bm, pl.splits);
      if (b > pl.ws_bytes) pl.ws_bytes = b;
    }
  }
#endif
  return pl;
}

template <int BM, bool AKF, bool BKF>
static void launch_tile(const M2dGemmParams& p, dim3 grid, hipStream_t stream) {
  const bool masked = p.A.mask || p.B.mask;
  if (masked && p.O.wide) hipLaunchKernelGGL((m2d_gemm_kernel<BM, 128, AKF, BKF, true, true>), grid, dim3(256), 0, stream, p);
  else if (masked) hipLaunchKernelGGL((m2d_gemm_kernel<BM, 128, AKF, BKF, true, false>), grid, dim3(256), 0, stream, p);
  else if (p.O.wide) hipLaunchKernelGGL((m2d_gemm_kernel<BM, 128, AKF, BKF, false, true>), grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((m2d_gemm_kernel<BM, 128, AKF, BKF, false, false>), grid, dim3(256), 0, stream, p);
}

// M2D_DL=0: keep row-fast / row-fast launches on the register-staging kernel (A/B lever)
static bool dl_enabled() {
  static const bool on = [] { const char* e = getenv("M2D_DL"); return !(e && e[0] == '0'); }();
  return on;
}
// (for callers that pick a weight packing only the LDS-direct kernels can read: conv1d.hip's phase-major sub-pixel form)
bool m2d_dl_enabled() { return dl_enabled(); }

// THE gate of the LDS-direct kernel: TileMap::load_lds implements row-fast operands with the one-level row / k maps
// only - no operand mask, no bias ("ones") row, no window views (rdiv2 / kdiv2). A new operand feature must be refused
// here (or taught to load_lds) before it can reach m2d_gemm_dl_kernel.
static inline bool dl_eligible(const M2dGemmParams& p, bool akf, bool bkf) {
  return !akf && !bkf && !p.A.mask && !p.B.mask && !p.A.ones_row_p1 && !p.B.ones_row_p1 && p.A.rdiv2 <= 0 &&
         p.B.rdiv2 <= 0 && p.A.kdiv2 <= 0 && p.B.kdiv2 <= 0;
}

// M2D_DL8=0: 128-row LDS-direct launches keep four waves per workgroup (A/B lever)
static bool dl8_enabled() {
  static const bool on = [] { const char* e = getenv("M2D_DL8"); return !(e && e[0] == '0'); }();
  return on;
}

template <int BM>
static int launch_maps(const M2dGemmParams& p, bool akf, bool bkf, dim3 grid, hipStream_t stream) {
  {
    if (dl_enabled() && dl_eligible(p, akf, bkf)) {
      if constexpr (BM == 128) {
        if (p.tall_last_rb) {
          hipLaunchKernelGGL(m2d_gemm_dl_tall_kernel, grid, dim3(256), 0, stream, p);
          return 0;
        }
        // eight waves per workgroup on the same tile (M2dTiling); the one-element epilogue does not fit the 64 VGPRs of
        // eight waves per SIMD without scratch: those launches (strided backward-data) keep four waves
        if (dl8_enabled() && (p.O.wide || (p.O.quad && p.splits <= 1))) {
          if (p.O.wide) hipLaunchKernelGGL((m2d_gemm_dl_kernel<128, 128, 1, 8>), grid, dim3(512), 0, stream, p);
          else hipLaunchKernelGGL((m2d_gemm_dl_kernel<128, 128, 2, 8>), grid, dim3(512), 0, stream, p);
          return 0;
        }
      }
      if (p.O.wide) hipLaunchKernelGGL((m2d_gemm_dl_kernel<BM, 128, 1>), grid, dim3(256), 0, stream, p);
      else if (p.O.quad && p.splits <= 1) hipLaunchKernelGGL((m2d_gemm_dl_kernel<BM, 128, 2>), grid, dim3(256), 0, stream, p);
      else hipLaunchKernelGGL((m2d_gemm_dl_kernel<BM, 128, 0>), grid, dim3(256), 0, stream, p);
      return 0;
    }
  }
  if (akf && !bkf) launch_tile<BM, true, false>(p, grid, stream);
  else if (!akf && !bkf) launch_tile<BM, false, false>(p, grid, stream);
  else if (akf && bkf) launch_tile<BM, true, true>(p, grid, stream);
  else return -1;
  return 0;
}

// ---- per-stream zero-kept scratch (arrival counters of the in-kernel split-K fix-up) ------------------------------
// The caller registers, once per stream, a zeroed device buffer it keeps alive (m2d_stream_scratch_set); launches on
// that stream take their tickets from it and leave them zero. Launches of one stream run in order, so one buffer per
// stream is enough; a stream without one uses the two-launch split-K.
#include <map>
#include <mutex>
static std::mutex g_scratch_mu;
static std::map<hipStream_t, std::pair<unsigned*, size_t>> g_scratch;

unsigned* m2d_stream_scratch_get(hipStream_t stream, size_t* bytes) {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  auto it = g_scratch.find(stream);
  if (it == g_scratch.end()) {
    if (bytes) *bytes = 0;
    return nullptr;
  }
  if (bytes) *bytes = it->second.second;
  return it->second.first;
}

extern "C" int m2d_stream_create(void** stream) {
  if (!stream) M2D_FAIL(M2D_ERR_ARG, "m2d_stream_create: null argument");
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) M2D_FAIL(M2D_ERR_HIP, "m2d_stream_create: hipStreamCreateWithFlags failed");
  *stream = (void*)s;
  return M2D_OK;
}

extern "C" int m2d_stream_scratch_set(void* stream, void* zeroed, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!zeroed || bytes == 0) g_scratch.erase((hipStream_t)stream);
  else g_scratch[(hipStream_t)stream] = std::make_pair((unsigned*)zeroed, bytes);
  return M2D_OK;
}

// split-K in one launch when the stream has tickets for every tile, the slab holds whole-tile images and stays under
// the buffer-addressing limit (M2D_FUSED_SPLITK=0: never)
static void decide_fused(M2dGemmParams& p, int bm, int splits, const void* ws, size_t ws_bytes, hipStream_t stream) {
  static const bool on = [] { const char* e = getenv("M2D_FUSED_SPLITK"); return !(e && e[0] == '0'); }();
  static const int max_splits = [] { const char* e = getenv("M2D_FUSE_MAX"); return e ? atoi(e) : M2D_FUSE_MAX_SPLITS; }();  // A/B lever
  p.tickets = nullptr;
  if (!on || splits <= 1 || splits > max_splits || p.bwd_data || !ws) return;
  const size_t need = m2d_slab_bytes(p.M, p.N, bm, splits);
  if (need > ws_bytes || need >= 0x7fffffffULL || ((uintptr_t)ws & 15u)) return;
  size_t sb = 0;
  unsigned* t = m2d_stream_scratch_get(stream, &sb);
  const size_t tiles = (size_t)m2d_ceil_div(p.M, bm) * (size_t)m2d_ceil_div(p.N, 128);
  if (!t || tiles * sizeof(unsigned) > sb) return;
  p.tickets = t;
}

// would decide_fused take this plan? (Launches with epilogue statistics may split K only in the one-launch form: the
// workgroup that draws a tile's last ticket holds the whole K and runs the ordinary epilogue, statistics included; the
// two-launch form's reduction kernel has no statistics pass.)
static bool fused_possible(const M2dGemmParams& p, int bm, int splits, const void* ws, size_t ws_bytes, hipStream_t stream) {
  M2dGemmParams q = p;
  decide_fused(q, bm, splits, ws, ws_bytes, stream);
  return q.tickets != nullptr;
}
static bool stats_split_enabled() {   // A/B lever
  static const bool on = [] { const char* e = getenv("M2D_STATS_SPLIT"); return !(e && e[0] == '0'); }();
  return on;
}

static int stats_narrow_fast_default() {   // A/B lever
  static const int v = [] { const char* e = getenv("M2D_STATS_NARROW_FAST"); return (e && e[0] == '0') ? 0 : 1; }();
  return v;
}
// M2D_TILE_MAP=0: tile id = workgroup id (A/B lever)
static int tile_map_default() {
  static const int v = [] { const char* e = getenv("M2D_TILE_MAP"); return e ? atoi(e) : 1; }();
  return v;
}

// one launch of the GEMM (+ the slab reduction) under plan (bm, splits)
static int plan_run(M2dGemmParams& p, int bm, int splits, bool a_kfast, bool b_kfast, void* ws, hipStream_t stream) {
  p.splits = splits;
  p.tile_map = tile_map_default();
  p.stats_narrow_fast = stats_narrow_fast_default();
  p.slab = splits > 1 ? (float*)ws : nullptr;
  const int mt = m2d_ceil_div(p.M, bm);
  const long long nt = m2d_ceil_div(p.N, 128);
  const dim3 grid((unsigned)nt, (unsigned)mt, (unsigned)(p.bwd_data ? p.phases : splits));
  int lrc;
  if (bm == 32) lrc = launch_maps<32>(p, a_kfast, b_kfast, grid, stream);
  else if (bm == 64) lrc = launch_maps<64>(p, a_kfast, b_kfast, grid, stream);
  else lrc = launch_maps<128>(p, a_kfast, b_kfast, grid, stream);
  if (lrc) return lrc;
  if (splits > 1 && !p.tickets) {
    const size_t total = (size_t)p.M * p.N;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(m2d_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
  }
  return 0;
}

// ---- plan cache: shape -> fastest timed plan --------------------------------------------------
#include <map>
#include <mutex>
#include <vector>
static std::mutex g_plan_mu;
static std::map<std::vector<int>, std::pair<int, int>> g_plan_cache;

static bool autotune_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("M2D_AUTOTUNE");
    on = (e && e[0] == '1') ? 1 : 0;
  }
  return on == 1;
}

// 16-byte epilogue rows (m2d_tile_epilogue, WIDE) when the output map allows it (M2D_WIDE_EPILOGUE=0: never)
static void decide_wide(M2dGemmParams& p, int splits, const void* ws) {
  static const bool wide_on = [] { const char* e = getenv("M2D_WIDE_EPILOGUE"); return !(e && e[0] == '0'); }();
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15u) == 0; };
  const M2dOutMap& o = p.O;
  bool w = wide_on && !p.bwd_data && (p.N % 4) == 0 && o.redirect_col_p1 == 0;
  if (splits > 1 && !p.tickets) {
    w = w && al16(ws);
  } else {
    w = w && o.m_div <= 0 && o.c_lo_stride == 1 && o.c_lim <= 0 && (o.cdiv % 4) == 0 && (o.c_hi_stride % 4) == 0 &&
        (o.m_stride % 4) == 0 && (o.c_off % 4) == 0 && (o.mask_wrap % 4) == 0 && al16(o.out) && al16(o.mask) && al16(o.residual) &&
        al16(o.sum_out) && (o.bias_mode != 2 || al16(o.bias));
  }
  p.O.wide = w ? 1 : 0;
}

int m2d_gemm_launch(M2dGemmParams& p, bool a_kfast, bool b_kfast, bool allow_split, void* ws,
                    size_t ws_bytes, hipStream_t stream, const char* what) {
  if (p.M <= 0 || p.N <= 0) return M2D_OK;
  if (p.A.nbytes == 0 || p.B.nbytes == 0)
    M2D_FAIL(M2D_ERR_RANGE, "%s: operand larger than 2 GiB (buffer addressing) or empty", what);
  if (p.M >= (1 << 24) || p.N >= (1 << 24))
    M2D_FAIL(M2D_ERR_RANGE, "%s: extent >= 2^24 (row decomposition is exact below that)", what);
  if (p.kdiv <= 0 || p.nhi < 0 || p.A.k_lo_stride < 0 || p.B.k_lo_stride < 0)
    M2D_FAIL(M2D_ERR_ARG, "%s: bad contraction map (kdiv=%d nhi=%d)", what, p.kdiv, p.nhi);
  if ((a_kfast && p.A.lim > 0) || (b_kfast && p.B.lim > 0))
    M2D_FAIL(M2D_ERR_ARG, "%s: k-fast operands carry no window", what);
  if (!((a_kfast && !b_kfast) || (!a_kfast && !b_kfast) || (a_kfast && b_kfast)))
    M2D_FAIL(M2D_ERR_ARG, "%s: unsupported operand map combination", what);
  if (p.phases < 1) p.phases = 1;
  {
    // the epilogue addresses out / mask / residual / sum_out with 32-bit byte offsets: the map must stay below 2 GiB
    const M2dOutMap& o = p.O;
    const long long row_off = o.m_div > 0 ? (long long)((p.M - 1) / o.m_div) * o.m_stride + (long long)(o.m_div - 1) * o.m_lo_stride
                                          : (long long)(p.M - 1) * o.m_stride;
    const long long col_off = p.bwd_data ? (long long)(p.ph_batch - 1) * o.c_hi_stride + p.ph_L
                                         : (long long)((p.N - 1) / (o.cdiv > 0 ? o.cdiv : 1)) * o.c_hi_stride +
                                               (long long)((o.cdiv > 0 ? o.cdiv : 1) - 1) * o.c_lo_stride + o.c_off;
    if ((row_off + col_off + 4) * 4 >= 0x7ffffff0LL) M2D_FAIL(M2D_ERR_RANGE, "%s: output larger than 2 GiB", what);
  }
  // bwd_data: the widest phase has ceil(ks / phases) taps
  const int nhi_max = p.bwd_data ? (p.ph_ks + p.phases - 1) / p.phases : p.nhi;
  const int nchunks = m2d_chunks(nhi_max, p.kdiv);
  // the statistics come out of the tile epilogue: split K only where the last arriver of a tile runs it (fused_possible)
  if (p.O.row_part && !stats_split_enabled()) allow_split = false;
  PlanCand cand[M2D_MAX_CAND];
  int nc = plan_candidates(p.M, p.N, nchunks, p.bwd_data ? p.phases : 1, allow_split, p.small_tile_penalty, cand, p.plan_kind);
  {
    // a split plan needs its slabs: drop the ones the workspace cannot hold
    int k = 0;
    for (int i = 0; i < nc; ++i) {
      const size_t need = cand[i].splits > 1 ? (size_t)cand[i].splits * (size_t)p.M * (size_t)p.N * sizeof(float) : 0;
      if (p.O.row_part && cand[i].splits > 1 && !fused_possible(p, cand[i].bm, cand[i].splits, ws, ws_bytes, stream)) continue;
      if (need == 0 || (ws != nullptr && ws_bytes >= need)) cand[k++] = cand[i];
    }
    if (k == 0 && p.O.row_part && allow_split) {   // every plan of the model splits K beyond the one-launch form: plan without
      nc = plan_candidates(p.M, p.N, nchunks, p.bwd_data ? p.phases : 1, false, p.small_tile_penalty, cand, p.plan_kind);
      k = nc;
    }
    if (k == 0) {
      if (nc > 0)
        M2D_FAIL(M2D_ERR_WORKSPACE, "%s: split-K needs %zu workspace bytes, got %zu", what,
                 (size_t)cand[0].splits * (size_t)p.M * (size_t)p.N * sizeof(float), ws_bytes);
      M2D_FAIL(M2D_ERR_ARG, "%s: no launch plan", what);
    }
    nc = k;
  }
  if (m2d_ceil_div(p.M, 32) > 65535) M2D_FAIL(M2D_ERR_RANGE, "%s: M too large (%d)", what, p.M);
  int bm = cand[0].bm, splits = cand[0].splits;
#ifdef M2D_TUNING
  const char* forced = getenv("M2D_PLAN");
  if (forced) {
    const M2dGemmPlan fp = m2d_gemm_plan(p.M, p.N, nchunks, p.bwd_data ? p.phases : 1, allow_split, p.small_tile_penalty, p.plan_kind);
    bm = fp.bm;
    splits = fp.splits;
    if (splits > 1 && (!ws || ws_bytes < (size_t)splits * p.M * p.N * sizeof(float)))
      M2D_FAIL(M2D_ERR_WORKSPACE, "%s: forced plan needs more workspace", what);
  } else
#endif
  if (autotune_enabled() && nc > 1 && !p.tall_last_rb) {   // (the phase-major launch has ONE plan: nothing to time)
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone;
    std::vector<int> key;
    key.reserve(40);
    for (const char* c = what; *c; ++c) key.push_back((int)*c);
    const int kf[] = {p.M, p.N, nchunks, p.bwd_data ? p.phases : 1, (int)a_kfast, (int)b_kfast,
                      (int)(p.A.mask || p.B.mask), (int)allow_split, p.lo_outer, p.kdiv, p.A.k_lo_stride,
                      p.B.k_lo_stride, p.B.r_lo_stride, (int)(p.small_tile_penalty * 100.f), nc,
                      cand[nc - 1].bm, cand[nc - 1].splits};
    key.insert(key.end(), kf, kf + sizeof(kf) / sizeof(int));
    std::unique_lock<std::mutex> lk(g_plan_mu);
    auto it = g_plan_cache.find(key);
    if (it != g_plan_cache.end()) {
      bm = it->second.first;
      splits = it->second.second;
    } else if (!capturing) {
      lk.unlock();
      // time every candidate on the real operands; two runs each, the faster one counts
      hipEvent_t e0, e1;
      if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        float best_ms = 1e30f;
        for (int i = 0; i < nc; ++i) {
          float ms = 1e30f;
          for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0, stream);
            if (plan_run(p, cand[i].bm, cand[i].splits, a_kfast, b_kfast, ws, stream)) break;
            (void)hipEventRecord(e1, stream);
            if (hipEventSynchronize(e1) != hipSuccess) break;
            float t = 0.f;
            if (hipEventElapsedTime(&t, e0, e1) == hipSuccess && t < ms) ms = t;
          }
          if (ms < best_ms) {
            best_ms = ms;
            bm = cand[i].bm;
            splits = cand[i].splits;
          }
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        M2D_CHECK_LAUNCH(what);
      }
      lk.lock();
      g_plan_cache[key] = std::make_pair(bm, splits);
    }
  }
  if (p.tall_last_rb) {  // the phase-major sub-pixel launch: whole 128-row tiles, whole K
    if (p.tall_last_rb != 1 || p.M % 128 != 0 || !p.lo_outer || p.O.bias_mode || p.O.act || !p.O.mask_last || !dl_enabled() ||
        !dl_eligible(p, a_kfast, b_kfast))
      M2D_FAIL(M2D_ERR_ARG, "%s: bad phase-major sub-pixel launch", what);
    bm = 128;
    splits = 1;
  }
#ifdef M2D_TUNING
  {
    const int lp[9] = {p.M, p.N, nchunks, p.bwd_data ? p.phases : 1, (int)allow_split, (int)(p.small_tile_penalty * 100.f), bm, splits,
                       p.plan_kind};
    memcpy(g_last_plan, lp, sizeof(lp));
  }
#endif
  decide_fused(p, bm, splits, ws, ws_bytes, stream);
  decide_wide(p, splits, ws);
  double flops = p.work_flops > 0.0 ? p.work_flops : 2.0 * p.M * (double)p.N * p.K;
  if (p.bwd_data) {
    flops = 0.0;
    for (int r = 0; r < p.phases; ++r) {
      const int taps = r < p.ph_ks ? (p.ph_ks - r + p.phases - 1) / p.phases : 0;
      const int qmin = r >= p.ph_pad ? 0 : (p.ph_pad - r + p.phases - 1) / p.phases;
      const int top = p.ph_L - 1 + p.ph_pad - r;
      const int nq = top >= 0 ? (top / p.phases - qmin + 1) : 0;
      if (nq > 0) flops += 2.0 * p.M * (double)p.ph_batch * nq * p.ph_cout * taps;
    }
  }
  {
    M2dProfScope prof(M2D_FAM_GEMM, stream, flops, 0.0, what, p.M, p.N, p.K);
    if (plan_run(p, bm, splits, a_kfast, b_kfast, ws, stream))
      M2D_FAIL(M2D_ERR_ARG, "%s: unsupported operand map combination", what);
    M2D_CHECK_LAUNCH(what);
    if (p.O.row_part) {
      if (!p.O.row_sums) M2D_FAIL(M2D_ERR_ARG, "%s: row statistics without a destination", what);
      // partials per row: one per wave column of every N tile; the 16-byte epilogue writes one per 32-column tile
      const int wn = 4;  // one per 32-column tile
      const int rc = m2d_rowsums_reduce(p.O.row_part, m2d_ceil_div(p.N, 128) * wn, p.M, p.O.row_sums,
                                        m2d_rowstats_scratch(p.O.row_part, p.M, p.N), stream);
      if (rc) return rc;
    }
  }
  return M2D_OK;
}

template <int BM>
static void k4_launch_tile(const M2dGemmParams& p, dim3 grid, hipStream_t stream) {
  if (p.O.wide) hipLaunchKernelGGL((m2d_conv_k4_kernel<BM, 128, true>), grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((m2d_conv_k4_kernel<BM, 128, false>), grid, dim3(256), 0, stream, p);
}

int m2d_conv_k4_launch(M2dGemmParams& p, bool allow_split, void* ws, size_t ws_bytes, hipStream_t stream,
                       const char* what) {
  if (p.M <= 0 || p.N <= 0) return M2D_OK;
  if (p.A.nbytes == 0 || p.B.nbytes == 0 || p.k4_ng <= 0)
    M2D_FAIL(M2D_ERR_RANGE, "%s: operand larger than 2 GiB (buffer addressing), empty, or no tap groups", what);
  if (p.M >= (1 << 24) || p.N >= (1 << 24)) M2D_FAIL(M2D_ERR_RANGE, "%s: extent >= 2^24", what);
  const int nchunks = (p.nhi * p.k4_ng + 3) / 4;
  if (p.O.row_part && !stats_split_enabled()) allow_split = false;
  PlanCand cand[M2D_MAX_CAND];
  const int nc = plan_candidates(p.M, p.N, nchunks, 1, allow_split, 1.0, cand);
  int bm = 128, splits = 1;
  for (int i = 0; i < nc; ++i) {
    const size_t need = cand[i].splits > 1 ? (size_t)cand[i].splits * (size_t)p.M * (size_t)p.N * sizeof(float) : 0;
    // (statistics: one-launch split-K only, see m2d_gemm_launch)
    if (p.O.row_part && cand[i].splits > 1 &&
        !fused_possible(p, cand[i].bm < 64 ? 64 : cand[i].bm, cand[i].splits, ws, ws_bytes, stream)) continue;
    if (need == 0 || (ws != nullptr && ws_bytes >= need)) {
      bm = cand[i].bm < 64 ? 64 : cand[i].bm;
      splits = cand[i].splits;
      break;
    }
  }
  p.splits = splits;
  p.tile_map = tile_map_default();
  p.stats_narrow_fast = stats_narrow_fast_default();
  p.slab = splits > 1 ? (float*)ws : nullptr;
  decide_fused(p, bm, splits, ws, ws_bytes, stream);
  decide_wide(p, splits, ws);
  const dim3 grid((unsigned)m2d_ceil_div(p.N, 128), (unsigned)m2d_ceil_div(p.M, bm), (unsigned)splits);
  {
    M2dProfScope prof(M2D_FAM_GEMM, stream, 2.0 * p.M * (double)p.N * p.K, 0.0, what, p.M, p.N, p.K);
    if (bm == 64) k4_launch_tile<64>(p, grid, stream);
    else k4_launch_tile<128>(p, grid, stream);
    if (splits > 1 && !p.tickets) {
      const size_t total = (size_t)p.M * p.N;
      unsigned blocks = (unsigned)((total + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(m2d_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    }
    M2D_CHECK_LAUNCH(what);
    if (p.O.row_part) {
      if (!p.O.row_sums) M2D_FAIL(M2D_ERR_ARG, "%s: row statistics without a destination", what);
      const int wn = 4;  // one per 32-column tile
      const int rc = m2d_rowsums_reduce(p.O.row_part, m2d_ceil_div(p.N, 128) * wn, p.M, p.O.row_sums,
                                        m2d_rowstats_scratch(p.O.row_part, p.M, p.N), stream);
      if (rc) return rc;
    }
  }
  return M2D_OK;
}

// number of operand shapes with a timed launch plan (diagnostics)
extern "C" int m2d_plan_cache_size(void) {
  std::lock_guard<std::mutex> lk(g_plan_mu);
  return (int)g_plan_cache.size();
}
