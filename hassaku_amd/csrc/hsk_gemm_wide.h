// hsk_gemm_wide.h -- the k loop of the 256 x 256 bf16x3 score GEMM (one wave per SIMD) of k_score_gemm_x3_wide
// (hsk_eval.hip).  Every output element is accumulated through the same sequence of MFMAs as in the 128 x 128 kernels, so
// the scores are the same bits.  (An in-GEMM selection on this core -- a k_score_topk with 256-column tiles -- was built,
// bit-equal, and measured SLOWER than the 128 x 128 one, 838 against 927 k users/s at the lfm2b shape: the selection's
// epilogue is ~12 000 instructions of compares and rare appends per tile and wave; with two workgroups per CU it runs
// under the other workgroup's MFMAs, with one wave per SIMD nothing covers it.  Not kept.)
//
// A workgroup of four waves owns 256 x 256 outputs, wave (wm, wn) the 128 x 128 block at (128 wm, 128 wn): 4 x 4
// accumulator tiles of v_mfma_f32_32x32x16_bf16 = 256 accumulator registers, 0.25 fragment reads per MFMA.  Operands are
// the bf16 pieces made by k_split_planes<16>: [k16-tile][row][piece][16], rows padded to 256 with zeros -- a block's
// pieces of one k-tile are 256 x 96 contiguous bytes = 1536 chunks of 16 bytes, six per thread and operand.  They go
// global -> registers -> LDS one k-step ahead; LDS holds two stages (rows of 48 bytes: ds_read_b128 conflict-free), one
// barrier per k-step.  A step is ONE hand-interleaved instruction stream pinned by scheduling fences -- a (store, load)
// pair after every 8 MFMAs, the fragment reads of term n+1 under the MFMAs of term n: as a burst at the head of the step
// the 12 stores + 12 loads held the wave, and its SIMD's matrix pipe, for ~1500 of the step's ~5000 cycles
// (profiles/probes/gemm_bf16x3_v4.hip: 52 -> 39 cycles per MFMA).
#pragma once
#include "hsk_common.h"

#define GEMM_W_BM 256
#define GEMM_W_BN 256
#define GEMM_W_BK 16
#define GEMM_W_LDK (GEMM_W_BK + 8)   // LDS row stride in bf16 elements
#define GEMM_W_A_STAGE (3 * GEMM_W_BM * GEMM_W_LDK)   // bf16 elements of one operand stage
#define GEMM_W_B_STAGE (3 * GEMM_W_BN * GEMM_W_LDK)
#define GEMM_W_LDS_BYTES (2 * (GEMM_W_A_STAGE + GEMM_W_B_STAGE) * 2)   // two stages of both operands: 147 456 bytes

typedef float hsk_w_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 hsk_w_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned hsk_w_u32x4 __attribute__((ext_vector_type(4)));   // (an array of HIP's uint4 structs ends up in scratch)

struct hsk_wide_stage {
  hsk_w_u32x4 ra[6], rb[6];   // the thread's twelve 16-byte chunks of the k-tile in flight
  int offa[6], offb[6];       // where they go inside an LDS stage (bf16 elements)
};

__device__ __forceinline__ void hsk_wide_init(hsk_wide_stage& s, int tid) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int c = tid + 256 * i, r = c / 6, j = c - r * 6;   // chunk c: row c / 6, piece (c % 6) / 2, half c % 2
    s.offa[i] = (j >> 1) * (GEMM_W_BM * GEMM_W_LDK) + r * GEMM_W_LDK + (j & 1) * 8;
    s.offb[i] = (j >> 1) * (GEMM_W_BN * GEMM_W_LDK) + r * GEMM_W_LDK + (j & 1) * 8;
  }
}
// a, b: the block's first byte inside one k-tile of the planes
__device__ __forceinline__ void hsk_wide_load(hsk_wide_stage& s, const __bf16* __restrict__ a,
                                              const __bf16* __restrict__ b, int tid) {
#pragma unroll
  for (int i = 0; i < 6; ++i) s.ra[i] = *reinterpret_cast<const hsk_w_u32x4*>(a + (tid + 256 * i) * 8);
#pragma unroll
  for (int i = 0; i < 6; ++i) s.rb[i] = *reinterpret_cast<const hsk_w_u32x4*>(b + (tid + 256 * i) * 8);
}
__device__ __forceinline__ void hsk_wide_store(const hsk_wide_stage& s, __bf16* as_stage, __bf16* bs_stage) {
#pragma unroll
  for (int i = 0; i < 6; ++i) *reinterpret_cast<hsk_w_u32x4*>(as_stage + s.offa[i]) = s.ra[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) *reinterpret_cast<hsk_w_u32x4*>(bs_stage + s.offb[i]) = s.rb[i];
}

// The k loop.  On entry: LDS stage 0 holds k-tile 0, the registers of `s` k-tile 1 (or tile 0 again when NT == 1), every
// wave is past the barrier behind those stores, acc is zero.  a0 / b0: the block's first byte in k-tile 0 of the planes,
// a_step / b_step: elements from one k-tile to the next (padded rows * 48).  On exit every wave is past the last
// barrier (the LDS stages are free).
__device__ __forceinline__ void hsk_wide_kloop(hsk_w_f32x16 (&acc)[4][4], hsk_wide_stage& s, __bf16* As, __bf16* Bs,
                                               const __bf16* __restrict__ a0, const __bf16* __restrict__ b0,
                                               long long a_step, long long b_step, int NT, int tid, int wm, int wn,
                                               int r32, int h) {
  constexpr int BM = GEMM_W_BM, BN = GEMM_W_BN, LDK = GEMM_W_LDK, TM = 4, TN = 4, WM = 128, WN = 128, CA = 6, CB = 6;
  for (int t = 0; t < NT; ++t) {
    const int buf = t & 1;
    // (unconditional stores / loads, the tile index clamped: the last steps re-load the last tile and store into a stage
    // nobody reads any more -- branches would cut the step into basic blocks)
    const __bf16* as = As + buf * GEMM_W_A_STAGE + (wm * WM + r32) * LDK + 8 * h;
    const __bf16* bs = Bs + buf * GEMM_W_B_STAGE + (wn * WN + r32) * LDK + 8 * h;
    const int tl = t + 2 < NT ? t + 2 : NT - 1;
    const __bf16* ga = a0 + (long long)tl * a_step;
    const __bf16* gbp = b0 + (long long)tl * b_step;
    __bf16* sa = As + (buf ^ 1) * GEMM_W_A_STAGE;
    __bf16* sb = Bs + (buf ^ 1) * GEMM_W_B_STAGE;
    hsk_w_bf16x8 a[3][TM], b[3][TN];
    auto read_a = [&](int pl) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[pl][i] = *reinterpret_cast<const hsk_w_bf16x8*>(as + pl * (BM * LDK) + i * 32 * LDK);
    };
    auto read_b = [&](int pl) {
#pragma unroll
      for (int j = 0; j < TN; ++j) b[pl][j] = *reinterpret_cast<const hsk_w_bf16x8*>(bs + pl * (BN * LDK) + j * 32 * LDK);
    };
    // the six products of weight >= 2^-16, smallest first: (3,1) (1,3) (2,2) (2,1) (1,2) (1,1)
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
    constexpr int PER = TM * TN, NPAIR = CA + CB, CH = 6 * PER / NPAIR;   // 8 MFMAs per (store, load) pair
    static_assert(6 * PER % NPAIR == 0 && PER % CH == 0, "chunking");
    read_a(TA[0]);
    read_b(TB[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NPAIR; ++c) {
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int m = c * CH + q, tt = m / PER, i = (m % PER) / TN, j = m % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[tt]][i], b[TB[tt]][j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (c == 0) read_a(TA[1]);             // term 1 needs a[0], b[2]; term 2 a[1], b[1]; terms 3..5 reuse
      if (c == 1) read_b(TB[1]);
      if (c == PER / CH) read_a(TA[2]);
      if (c == PER / CH + 1) read_b(TB[2]);
      if (c < CA) {
        *reinterpret_cast<hsk_w_u32x4*>(sa + s.offa[c]) = s.ra[c];
        s.ra[c] = *reinterpret_cast<const hsk_w_u32x4*>(ga + (tid + 256 * c) * 8);
      } else {
        *reinterpret_cast<hsk_w_u32x4*>(sb + s.offb[c - CA]) = s.rb[c - CA];
        s.rb[c - CA] = *reinterpret_cast<const hsk_w_u32x4*>(gbp + (tid + 256 * (c - CA)) * 8);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
}
