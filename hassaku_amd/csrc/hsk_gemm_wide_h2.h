// hsk_gemm_wide_h2.h -- the k loop of the 256 x 256 score GEMM on TWO fp16 pieces per operand, three products per block.
//
// x (fp32), scaled by a power of two s so that the table's largest |x| sits in [2^14, 2^15), is cut into hi = fp16(s x)
// and lo = fp16(s x - hi): 11 + 11 significant bits, each cut exact in fp32; lo is a normal fp16 for every element within
// 2^-18 of the table's maximum (below that it loses bits it no longer matters to keep: |error| <= 2^-40 of the maximum).
// Of the four products per (a, b) pair the three of weight >= 2^-11 are kept -- a_lo b_hi, a_hi b_lo, a_hi b_hi, each a
// v_mfma_f32_32x32x16_f16 into the same fp32 accumulator (an fp16 x fp16 product is exact in fp32); the dropped a_lo b_lo
// and the pieces' own rounding are <= 2^-21 |a||b| per term -- measured below an fp32 GEMM's own rounding noise
// (profiles/probes/gemm_f16x2.hip) -- at HALF the MFMAs of the three-piece bf16 form (hsk_gemm_wide.h) and 4 instead of 6
// bytes per operand element through the L2s.  score = acc * 2^-(e_a + e_b), exact.
//
// Same geometry as hsk_gemm_wide.h: a workgroup of four waves (one per SIMD) owns 256 x 256 outputs, wave (wm, wn) the
// 128 x 128 block at (128 wm, 128 wn) in 4 x 4 accumulator tiles.  A k-step is 32 deep: two k16 tiles x two pieces =
// four 8 KB images [256 rows][32 B] per operand and stage, rows unpadded, the 16-byte half of a row XOR-ed with bit 4 of
// the row so that the four lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...) each cover all
// 64 banks.  LDS: 2 stages x 2 operands x 32 KB = 128 KB.  The pieces arrive from k_split_planes_h2 as
// [k16 tile][piece][padded row][16]: one image is 8 KB of consecutive bytes, eight 16-byte chunks per thread, operand
// and step; global -> registers -> LDS one step ahead; one barrier per step (96 MFMAs per wave).  As in the bf16 loop
// the step is ONE instruction stream pinned by scheduling fences: after every 6 MFMAs two fragment reads, one LDS
// store and one global load.
#pragma once
#include "hsk_gemm_wide.h"

typedef _Float16 hsk_w_f16x8 __attribute__((ext_vector_type(8)));

#define GEMM_H_BK 32
#define GEMM_H_IMAGE (256 * 32)                  // bytes: one (k16 tile, piece) image of 256 rows
#define GEMM_H_OP_STAGE (4 * GEMM_H_IMAGE)       // one operand, one stage: 32 KB
#define GEMM_H_STAGE (2 * GEMM_H_OP_STAGE)       // A then B
#define GEMM_H_LDS_BYTES (2 * GEMM_H_STAGE)      // 131 072

// exponent e of the scale 2^e for a table whose largest finite |x| is amax: 2^e amax in [2^14, 2^15); clamped so that
// 2^-(e_a + e_b) stays a normal fp32
__device__ __forceinline__ int hsk_h2_scale_exp(float amax) {
  if (!(amax > 0.f)) return 0;
  int e = 14 - ilogbf(amax);
  return e < -60 ? -60 : (e > 60 ? 60 : e);
}
// x -> (hi, lo) at scale 2^e.  Non-finite x (and a finite x whose scaled value overflows fp16): hi carries it, lo = 0.
__device__ __forceinline__ void hsk_split_h2(float x, int e, _Float16& hi, _Float16& lo) {
  const float xs = ldexpf(x, e);
  hi = (_Float16)xs;
  const float hf = (float)hi;
  lo = __builtin_isfinite(hf) ? (_Float16)(xs - hf) : (_Float16)0.f;
}

struct hsk_h2_stage {
  hsk_w_u32x4 ra[8], rb[8];   // the thread's sixteen 16-byte chunks of the k-step in flight
};

// chunk k (image k >> 1 = 2 * k16 + piece, half-image k & 1) of k-step T inside the planes of an operand whose padded row
// count is `rows`, for the block starting at row r0: a wave-uniform address (scalar registers) + the thread's 32-bit offset
__device__ __forceinline__ const hsk_w_u32x4* hsk_h2_src(const _Float16* __restrict__ P, int T, int k, int rows, int r0,
                                                         int tid) {
  const char* img = reinterpret_cast<const char*>(P) + ((long long)(4 * T + (k >> 1)) * rows + r0) * 32;   // uniform
  return reinterpret_cast<const hsk_w_u32x4*>(img + (unsigned)(tid * 16 + (k & 1) * 4096));
}
// byte offset of the thread's chunk k inside an operand stage
__device__ __forceinline__ int hsk_h2_dst(int k, int tid) {
  return (k >> 1) * GEMM_H_IMAGE + (k & 1) * 4096 + (tid >> 1) * 32 + (((tid & 1) ^ ((tid >> 5) & 1)) << 4);
}

__device__ __forceinline__ void hsk_h2_load(hsk_h2_stage& s, const _Float16* __restrict__ A, const _Float16* __restrict__ B,
                                            int a_rows, int b_rows, int m0, int n0, int T, int tid) {
#pragma unroll
  for (int k = 0; k < 8; ++k) s.ra[k] = *hsk_h2_src(A, T, k, a_rows, m0, tid);
#pragma unroll
  for (int k = 0; k < 8; ++k) s.rb[k] = *hsk_h2_src(B, T, k, b_rows, n0, tid);
}
__device__ __forceinline__ void hsk_h2_store(const hsk_h2_stage& s, unsigned char* stage, int tid) {
#pragma unroll
  for (int k = 0; k < 8; ++k) *reinterpret_cast<hsk_w_u32x4*>(stage + hsk_h2_dst(k, tid)) = s.ra[k];
#pragma unroll
  for (int k = 0; k < 8; ++k) *reinterpret_cast<hsk_w_u32x4*>(stage + GEMM_H_OP_STAGE + hsk_h2_dst(k, tid)) = s.rb[k];
}

// The k loop.  On entry: LDS stage 0 holds k-step 0, the registers of `s` k-step 1 (step 0 again when NT == 1), every wave
// is past the barrier behind those stores, acc is zero.  NT = Dp / 32.  On exit every wave is past the last barrier.
__device__ __forceinline__ void hsk_h2_kloop(hsk_w_f32x16 (&acc)[4][4], hsk_h2_stage& s, unsigned char* lds,
                                             const _Float16* __restrict__ A, const _Float16* __restrict__ B, int a_rows,
                                             int b_rows, int m0, int n0, int NT, int tid, int wm, int wn, int r32, int h) {
  constexpr int TM = 4, TN = 4, PER = TM * TN, NPAIR = 16, CH = 6 * PER / NPAIR;   // 6 MFMAs per (store, load) pair
  static_assert(CH * NPAIR == 6 * PER, "chunking");
  const int sw = (h ^ (r32 >> 4)) << 4;
  const int la = (wm * 128 + r32) * 32 + sw;                        // this lane's fragment bytes inside an A image
  const int lb = GEMM_H_OP_STAGE + (wn * 128 + r32) * 32 + sw;      // ... inside a B image
  for (int t = 0; t < NT; ++t) {
    const unsigned char* rd = lds + (t & 1) * GEMM_H_STAGE;
    unsigned char* wr = lds + ((t & 1) ^ 1) * GEMM_H_STAGE;
    const int tl = t + 2 < NT ? t + 2 : NT - 1;   // (clamped, unconditional: the last steps re-load the last tile)
    hsk_w_f16x8 a[2][2][TM], b[2][2][TN];         // [k16][piece][tile]
    auto read_a = [&](int k16, int pc, int lo, int hi) {
#pragma unroll
      for (int i = lo; i < hi; ++i)
        a[k16][pc][i] = *reinterpret_cast<const hsk_w_f16x8*>(rd + la + (2 * k16 + pc) * GEMM_H_IMAGE + i * 1024);
    };
    auto read_b = [&](int k16, int pc, int lo, int hi) {
#pragma unroll
      for (int j = lo; j < hi; ++j)
        b[k16][pc][j] = *reinterpret_cast<const hsk_w_f16x8*>(rd + lb + (2 * k16 + pc) * GEMM_H_IMAGE + j * 1024);
    };
    // per k16 tile the three products of weight >= 2^-11, smallest first: (lo, hi) (hi, lo) (hi, hi)
    constexpr int TK[6] = {0, 0, 0, 1, 1, 1}, TA[6] = {1, 0, 0, 1, 0, 0}, TB[6] = {0, 1, 0, 0, 1, 0};
    read_a(0, 1, 0, 1);   // (in the order the first MFMAs want them: the first one waits for two reads, not five)
    read_b(0, 0, 0, 4);
    read_a(0, 1, 1, 4);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NPAIR; ++c) {
      // fragments of the coming terms, the store of the next step's chunk and the load of the one after, each behind an
      // MFMA of its own: issued as one burst behind the chunk's six MFMAs they drained the matrix pipe's queue
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int m = c * CH + q, tt = m / PER, i = (m % PER) / TN, j = m % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[TK[tt]][TA[tt]][i], b[TK[tt]][TB[tt]][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (q == 0) {
          if (c < 8) *reinterpret_cast<hsk_w_u32x4*>(wr + hsk_h2_dst(c, tid)) = s.ra[c];
          else *reinterpret_cast<hsk_w_u32x4*>(wr + GEMM_H_OP_STAGE + hsk_h2_dst(c - 8, tid)) = s.rb[c - 8];
        }
        if (q == 1) {
          if (c < 8) s.ra[c] = *hsk_h2_src(A, tl, c, a_rows, m0, tid);
          else s.rb[c - 8] = *hsk_h2_src(B, tl, c - 8, b_rows, n0, tid);
        }
        if (q == 2 || q == 3) {
          const int lo = 2 * (q - 2);   // two fragment reads behind this MFMA
          if (c == 0) read_a(0, 0, lo, lo + 2);
          if (c == 1) read_b(0, 1, lo, lo + 2);
          if (q == 2) {
            if (c == 3) read_a(1, 1, 0, 2);
            if (c == 4) read_a(1, 1, 2, 4);
            if (c == 5) read_b(1, 0, 0, 2);
            if (c == 6) read_b(1, 0, 2, 4);
            if (c == 7) read_b(1, 1, 0, 2);
            if (c == 8) read_b(1, 1, 2, 4);
            if (c == 9) read_a(1, 0, 0, 2);
            if (c == 10) read_a(1, 0, 2, 4);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
}
