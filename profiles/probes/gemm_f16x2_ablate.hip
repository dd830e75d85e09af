// Probe (not part of the product): what each part of the fp16-pair k loop (csrc/hsk_gemm_wide_h2.h, copied below with switches)
// costs -- global loads + LDS stores, fragment reads, the barrier -- on CONSTANT operands (which clock higher than random
// ones: compare lines of this probe with each other only), and what a ragged last round of workgroups costs.
//   hipcc -O3 --offload-arch=gfx950 -I../../hassaku_amd/csrc -I../../include gemm_f16x2_ablate.hip -o abl && ./abl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "hsk_gemm_wide_h2.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int ABL>
__device__ __forceinline__ void abl_kloop(hsk_w_f32x16 (&acc)[4][4], hsk_h2_stage& s, unsigned char* lds,
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
    hsk_w_f16x8 a[2][2][TM], b[2][2][TN];
    if (ABL & 2) {
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int z = 0; z < 4; ++z)
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[x][y][z][e] = (_Float16)(float)(tid + e + z); b[x][y][z][e] = (_Float16)(float)(tid - e - y); }
    }
    auto read_a = [&](int k16, int pc, int lo, int hi) {
#pragma unroll
      for (int i = lo; i < hi; ++i)
        if (!(ABL & 2)) a[k16][pc][i] = *reinterpret_cast<const hsk_w_f16x8*>(rd + la + (2 * k16 + pc) * GEMM_H_IMAGE + i * 1024);
    };
    auto read_b = [&](int k16, int pc, int lo, int hi) {
#pragma unroll
      for (int j = lo; j < hi; ++j)
        if (!(ABL & 2)) b[k16][pc][j] = *reinterpret_cast<const hsk_w_f16x8*>(rd + lb + (2 * k16 + pc) * GEMM_H_IMAGE + j * 1024);
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
          if (!(ABL & 1)) { if (c < 8) *reinterpret_cast<hsk_w_u32x4*>(wr + hsk_h2_dst(c, tid)) = s.ra[c];
          else *reinterpret_cast<hsk_w_u32x4*>(wr + GEMM_H_OP_STAGE + hsk_h2_dst(c - 8, tid)) = s.rb[c - 8]; }
        }
        if (q == 1) {
          if (!(ABL & 1)) { if (c < 8) s.ra[c] = *hsk_h2_src(A, tl, c, a_rows, m0, tid);
          else s.rb[c - 8] = *hsk_h2_src(B, tl, c - 8, b_rows, n0, tid); }
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
    if (!(ABL & 4)) __syncthreads();
  }
}

template <int ABL>
__global__ __launch_bounds__(256, 1) void k_abl(const _Float16* __restrict__ Apl, const _Float16* __restrict__ Bpl, float* __restrict__ C, int a_rows, int b_rows, int kdim) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hlds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r32 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 256, n0 = blockIdx.x * 256;
  hsk_w_f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  hsk_h2_stage stg;
  const int NT = kdim / 32;
  hsk_h2_load(stg, Apl, Bpl, a_rows, b_rows, m0, n0, 0, tid);
  hsk_h2_store(stg, hlds, tid);
  hsk_h2_load(stg, Apl, Bpl, a_rows, b_rows, m0, n0, NT > 1 ? 1 : 0, tid);
  __syncthreads();
  abl_kloop<ABL>(acc, stg, hlds, Apl, Bpl, a_rows, b_rows, m0, n0, NT, tid, wm, wn, r32, h);
  float s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
  if (s == 12345.678f) C[0] = s;
}
template <int ABL> void run(const char* what, _Float16* A, _Float16* B, float* C, int M, int N, int K) {
  CK(hipFuncSetAttribute((const void*)k_abl<ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_H_LDS_BYTES));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float tot = 0;
  for (int rep = 0; rep < 22; ++rep) {
    CK(hipEventRecord(e0)); k_abl<ABL><<<dim3(N / 256, M / 256), 256, GEMM_H_LDS_BYTES>>>(A, B, C, M, N, K); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 2) tot += ms;
  }
  printf("%-44s %.1f us  (%.0f TF of f16 MFMA)\n", what, tot / 20 * 1e3, 6.0 * M * N * K / (tot / 20 * 1e-3) / 1e12);
}
int main() {
  const int M = 8192, N = 10752 + 256 * 0, K = 512;   // 32 x 42 = 1344 workgroups: 5.25 rounds
  _Float16 *A, *B; float* C;
  CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, 1 << 20));
  CK(hipMemset(A, 0x3c, (size_t)M * K * 4)); CK(hipMemset(B, 0x3c, (size_t)N * K * 4));
  run<0>("full k loop", A, B, C, M, N, K);
  run<1>("no global loads / LDS stores", A, B, C, M, N, K);
  run<2>("no fragment reads", A, B, C, M, N, K);
  run<3>("neither (MFMAs + barrier)", A, B, C, M, N, K);
  run<7>("MFMAs only (no barrier either)", A, B, C, M, N, K);
  run<4>("full, no barrier (wrong results)", A, B, C, M, N, K);
  // a grid that is a whole number of rounds: 32 x 40 = 1280 = 5 x 256
  run<0>("full k loop, 1280 workgroups (5 rounds)", A, B, C, M, 10240, K);
  return 0;
}
