// Probe (not part of the product): the bf16x3 score GEMM with a 256 x 256 block tile, ONE wave per SIMD.
//
// The 128 x 128 x 32 tiling of k_score_gemm_x3 (4 waves of 64 x 64, two workgroups per CU) moves 6 bytes per operand
// element through the L2s for 128 flop/B: at the bf16 MFMA peak that is 19.5 TB/s -- above what the L2s deliver -- and
// its two workgroups per CU share every SIMD's matrix pipe.  Here a workgroup owns 256 x 256 (256 flop/B: 9.8 TB/s at
// peak), its four waves own 128 x 128 each (4 x 4 accumulator tiles = 256 registers, 0.25 fragment reads per MFMA) and
// run alone on their SIMDs with the whole 512-register file; the pieces ([k16-tile][row][piece][16] bf16, made once by a
// pre-pass) go global -> registers -> LDS two k-steps ahead, LDS double-buffered at BK = 16, one barrier per k-step.
// Same accumulation order per output element as the product kernel: bit-identical scores.
//   hipcc -O3 --offload-arch=gfx950 gemm_bf16x3_v4.hip -o gemm_bf16x3_v4 && ./gemm_bf16x3_v4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int M = 8192, N = 10752, K = 512;
constexpr int BK = 16, LDK = BK + 8;   // 48-byte LDS rows: ds_read_b128 conflict-free
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  l = (__bf16)r2;
}

// pieces of X [rows, K] -> P [K/16][rows][3][16]
__global__ __launch_bounds__(256) void k_presplit16(const float* __restrict__ X, int rows, __bf16* __restrict__ P) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int per_row = K / 4;
  const long long r = t / per_row;
  const int k = (int)(t - r * per_row) * 4;
  if (r >= rows) return;
  const float4 x = *reinterpret_cast<const float4*>(X + r * K + k);
  const float v[4] = {x.x, x.y, x.z, x.w};
  bf16x4 p1, p2, p3;
  for (int e = 0; e < 4; ++e) { __bf16 a, b, c; split3(v[e], a, b, c); p1[e] = a; p2[e] = b; p3[e] = c; }
  __bf16* dst = P + ((long long)(k / BK) * rows + r) * 48 + (k % BK);
  *reinterpret_cast<bf16x4*>(dst) = p1;
  *reinterpret_cast<bf16x4*>(dst + 16) = p2;
  *reinterpret_cast<bf16x4*>(dst + 32) = p3;
}

template <int BM, int BN, int WM, int WN, bool PROF = false, int OCC = 1>   // block tile, per-wave tile; (BM / WM) * (BN / WN) == 4 waves; PROF: diagnostic stamps
__global__ __launch_bounds__(256, OCC) void k_gemm_v4(const __bf16* __restrict__ Ap, const __bf16* __restrict__ Bp,
                                                     float* __restrict__ C, int m_rows, int n_rows,
                                                     unsigned long long* __restrict__ stamps_ = nullptr) {
  unsigned long long* stamps = PROF ? stamps_ : nullptr;
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_STAGE = 3 * BM * LDK, B_STAGE = 3 * BN * LDK;   // bf16 elements per stage
  __bf16* As = lds;                    // [2][3][BM * LDK]
  __bf16* Bs = lds + 2 * A_STAGE;      // [2][3][BN * LDK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int WAVES_N = BN / WN;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int r32 = lane & 31, h = lane >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  // staging: a tile's pieces are BM * 96 bytes of consecutive global memory = BM * 6 chunks of 16 bytes
  constexpr int CA = BM * 6 / 256, CB = BN * 6 / 256;
  u32x4 ra[CA], rb[CB];
  int offa[CA], offb[CB];   // LDS element offset of each chunk inside a stage
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int c = tid + 256 * i, r = c / 6, j = c - r * 6;
    offa[i] = (j >> 1) * (BM * LDK) + r * LDK + (j & 1) * 8;
  }
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int c = tid + 256 * i, r = c / 6, j = c - r * 6;
    offb[i] = (j >> 1) * (BN * LDK) + r * LDK + (j & 1) * 8;
  }
  auto load_tile = [&](int t, u32x4 (&qa)[CA], u32x4 (&qb)[CB]) {
    const __bf16* a = Ap + ((long long)t * m_rows + m0) * 48;
    const __bf16* b = Bp + ((long long)t * n_rows + n0) * 48;
#pragma unroll
    for (int i = 0; i < CA; ++i) qa[i] = *reinterpret_cast<const u32x4*>(a + (tid + 256 * i) * 8);
#pragma unroll
    for (int i = 0; i < CB; ++i) qb[i] = *reinterpret_cast<const u32x4*>(b + (tid + 256 * i) * 8);
  };
  auto store_tile = [&](int buf, const u32x4 (&qa)[CA], const u32x4 (&qb)[CB]) {
#pragma unroll
    for (int i = 0; i < CA; ++i) *reinterpret_cast<u32x4*>(As + buf * A_STAGE + offa[i]) = qa[i];
#pragma unroll
    for (int i = 0; i < CB; ++i) *reinterpret_cast<u32x4*>(Bs + buf * B_STAGE + offb[i]) = qb[i];
  };
  constexpr int NT = K / BK;
  load_tile(0, ra, rb);
  store_tile(0, ra, rb);
  load_tile(1, ra, rb);
  __syncthreads();
  // diagnostic build only (stamps != NULL): shader clock = d(s_memtime) / d(s_memrealtime) x 100 MHz around the loop
  unsigned long long t_c0 = 0, t_r0 = 0;
  if (stamps) { t_c0 = __builtin_amdgcn_s_memtime(); t_r0 = __builtin_amdgcn_s_memrealtime(); }
  unsigned long long seg[3] = {0, 0, 0};   // diagnostic: cycles in [stores + load issue | fragment reads + MFMAs | barrier]
  for (int t = 0; t < NT; ++t) {
    const int buf = t & 1;
    // tile t+1 (loaded a step ago) -> LDS buffer buf^1 (last read in step t-1); its registers then take tile t+2.
    // Program order only: the scheduling pipeline below spreads these stores and loads over the step's MFMAs.
    // The step as ONE hand-interleaved stream, pinned by scheduling fences: a burst of 12 stores + 12 loads at the head
    // of the step holds the wave -- and its SIMD's matrix pipe -- for ~1500 cycles (measured: the CU's L1 takes 64 B per
    // clock, the LDS store path 13 cycles per 16-byte store); one (store, load) pair after every few MFMAs hides.  The
    // fragment reads of term n+1 ride under the MFMAs of term n.  (Unconditional stores / loads, the tile index clamped:
    // the last steps re-load the last tile and store into a buffer nobody reads any more -- branches would cut the
    // step into basic blocks.)
    const __bf16* as = As + buf * A_STAGE + (wm * WM + r32) * LDK + 8 * h;
    const __bf16* bs = Bs + buf * B_STAGE + (wn * WN + r32) * LDK + 8 * h;
    const int tl = t + 2 < NT ? t + 2 : NT - 1;
    const __bf16* ga = Ap + ((long long)tl * m_rows + m0) * 48;
    const __bf16* gb = Bp + ((long long)tl * n_rows + n0) * 48;
    __bf16* sa = As + (buf ^ 1) * A_STAGE;
    __bf16* sb = Bs + (buf ^ 1) * B_STAGE;
    bf16x8 a[3][TM], b[3][TN];
    auto read_a = [&](int pl) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[pl][i] = *reinterpret_cast<const bf16x8*>(as + pl * (BM * LDK) + i * 32 * LDK);
    };
    auto read_b = [&](int pl) {
#pragma unroll
      for (int j = 0; j < TN; ++j) b[pl][j] = *reinterpret_cast<const bf16x8*>(bs + pl * (BN * LDK) + j * 32 * LDK);
    };
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
    constexpr int PER = TM * TN, TOTAL = 6 * PER, NPAIR = CA + CB, CH = TOTAL / NPAIR;   // MFMAs per (store, load) pair
    static_assert(TOTAL % NPAIR == 0 && PER % CH == 0, "chunking");
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
      // fragments of the following terms: term 1 needs a[0], b[2]; term 2 needs a[1], b[1] (terms 3..5 reuse)
      {   // term k (k = 1, 2) starts at chunk k * PER / CH: its fragments are read behind the one or two chunks in front
        constexpr int S1 = PER / CH, S2 = 2 * PER / CH;
        if (c == (S1 >= 2 ? S1 - 2 : S1 - 1)) read_a(TA[1]);
        if (c == S1 - 1) read_b(TB[1]);
        if (c == (S1 >= 2 ? S2 - 2 : S2 - 1)) read_a(TA[2]);
        if (c == S2 - 1) read_b(TB[2]);
      }
      if (c < CA) {
        *reinterpret_cast<u32x4*>(sa + offa[c]) = ra[c];
        ra[c] = *reinterpret_cast<const u32x4*>(ga + (tid + 256 * c) * 8);
      } else {
        *reinterpret_cast<u32x4*>(sb + offb[c - CA]) = rb[c - CA];
        rb[c - CA] = *reinterpret_cast<const u32x4*>(gb + (tid + 256 * (c - CA)) * 8);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  if (stamps && (tid & 63) == 0) {
    const unsigned long long wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (tid == 0) {
      stamps[8 * wg] = __builtin_amdgcn_s_memtime() - t_c0;
      stamps[8 * wg + 1] = __builtin_amdgcn_s_memrealtime() - t_r0;
      stamps[8 * wg + 2] = seg[0]; stamps[8 * wg + 3] = seg[1]; stamps[8 * wg + 4] = seg[2];
    }
    if (tid == 192) { stamps[8 * wg + 5] = seg[0]; stamps[8 * wg + 6] = seg[1]; stamps[8 * wg + 7] = seg[2]; }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + wm * WM + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        const int col = n0 + wn * WN + j * 32 + r32;
        C[(long long)row * N + col] = acc[i][j][q];
      }
}

template <int BM, int BN, int WM, int WN, int OCC = 1>
static void run(const char* name, const __bf16* Ap, const __bf16* Bp, float* C, const float* Cref_dev) {
  const size_t lds = (size_t)2 * 3 * (BM + BN) * LDK * 2;
  CK(hipFuncSetAttribute((const void*)k_gemm_v4<BM, BN, WM, WN, false, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  dim3 grid(N / BN, M / BM);
  float tot = 0, best = 1e9;
  for (int rep = 0; rep < 22; ++rep) {
    CK(hipEventRecord(e0));
    k_gemm_v4<BM, BN, WM, WN, false, OCC><<<grid, 256, lds>>>(Ap, Bp, C, M, N);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 2) { tot += ms; best = fminf(best, ms); }
  }
  CK(hipGetLastError());
  const double flop = 2.0 * M * N * K;
  std::vector<float> c1((size_t)256 * N), c2((size_t)256 * N);
  CK(hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(c2.data(), Cref_dev, c2.size() * 4, hipMemcpyDeviceToHost));
  size_t diff = 0; for (size_t i = 0; i < c1.size(); ++i) diff += c1[i] != c2[i];
  printf("%-28s mean %.1f us best %.1f us = %.1f TFLOP/s fp32-equivalent (%.2f of the six-product bf16 peak), LDS %zu B, %zu values differ from the 128x128 kernel\n",
         name, tot / 20 * 1e3, best * 1e3, flop / (tot / 20 * 1e-3) / 1e12, flop * 6 / (tot / 20 * 1e-3) / 2.5e15, lds, diff);
  {   // diagnostic launches: the clock the chip holds inside the loop, and the loop's cycles per MFMA
    const int n_wg = (N / BN) * (M / BM);
    unsigned long long* st; CK(hipMalloc(&st, (size_t)n_wg * 64));
    CK(hipFuncSetAttribute((const void*)k_gemm_v4<BM, BN, WM, WN, true, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 30; ++rep) k_gemm_v4<BM, BN, WM, WN, true, OCC><<<grid, 256, lds>>>(Ap, Bp, C, M, N, st);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs((size_t)n_wg * 8);
    CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    double sg[6] = {0, 0, 0, 0, 0, 0};
    for (int w = 0; w < n_wg; ++w) if (hs[8 * w + 1] > 0) { clk.push_back((double)hs[8 * w] / (double)hs[8 * w + 1] * 0.1); cyc.push_back((double)hs[8 * w]); for (int q = 0; q < 6; ++q) sg[q] += (double)hs[8 * w + 2 + q] / n_wg / (K / BK); }

    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double mfma_per_wave = 6.0 * (WM / 32) * (WN / 32) * (K / BK);
    printf("    in-kernel clock (median over workgroups) %.2f GHz; loop = %.0f shader cycles = %.1f cycles per MFMA of a wave (32 = the matrix pipe's pace)\n",
           clk[clk.size() / 2], cyc[cyc.size() / 2], cyc[cyc.size() / 2] / mfma_per_wave);
    CK(hipFree(st));
  }
}

// reference: the 128 x 128 x 16 double-buffered kernel of gemm_bf16x3.hip (k_gemm_planes2), as the baseline
__global__ __launch_bounds__(256, 2) void k_gemm_base(const __bf16* __restrict__ Ap, const __bf16* __restrict__ Bp, float* __restrict__ C) {
  constexpr int BM = 128, BN = 128;
  __shared__ __attribute__((aligned(16))) __bf16 As[2][3][BM * LDK];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2][3][BN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int r32 = lane & 31, h = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  u32x4 ra[3], rb[3];
  int lds_off[3], lds_pl[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = tid + 256 * i, r = c / 6, j = c - r * 6;
    lds_pl[i] = j >> 1;
    lds_off[i] = r * LDK + (j & 1) * 8;
  }
  auto load_tile = [&](int t) {
    const __bf16* a = Ap + ((long long)t * M + m0) * 48;
    const __bf16* b = Bp + ((long long)t * N + n0) * 48;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      ra[i] = *reinterpret_cast<const u32x4*>(a + (tid + 256 * i) * 8);
      rb[i] = *reinterpret_cast<const u32x4*>(b + (tid + 256 * i) * 8);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      *reinterpret_cast<u32x4*>(&As[buf][lds_pl[i]][lds_off[i]]) = ra[i];
      *reinterpret_cast<u32x4*>(&Bs[buf][lds_pl[i]][lds_off[i]]) = rb[i];
    }
  };
  constexpr int NT = K / BK;
  load_tile(0);
  store_tile(0);
  load_tile(1);
  __syncthreads();
  for (int t = 0; t < NT; ++t) {
    const int buf = t & 1;
    if (t + 1 < NT) store_tile(buf ^ 1);
    if (t + 2 < NT) load_tile(t + 2);
    bf16x8 a[3][2], b[3][2];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[pl][i] = *reinterpret_cast<const bf16x8*>(&As[buf][pl][(wm * 64 + i * 32 + r32) * LDK + 8 * h]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[pl][j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][pl][(wn * 64 + j * 32 + r32) * LDK + 8 * h]);
    }
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int tt = 0; tt < 6; ++tt)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[tt]][i], b[TB[tt]][j], acc[i][j], 0, 0, 0);
    __syncthreads();
  }
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        const int col = n0 + wn * 64 + j * 32 + r32;
        C[(long long)row * N + col] = acc[i][j][q];
      }
}

int main() {
  float *A, *B, *C, *Cref;
  CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4));
  CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&Cref, (size_t)M * N * 4));
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  srand(3);
  for (auto& v : hA) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  for (auto& v : hB) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  __bf16 *Ap, *Bp;
  CK(hipMalloc(&Ap, (size_t)M * K * 6)); CK(hipMalloc(&Bp, (size_t)N * K * 6));
  k_presplit16<<<(unsigned)(((size_t)M * K / 4 + 255) / 256), 256>>>(A, M, Ap);
  k_presplit16<<<(unsigned)(((size_t)N * K / 4 + 255) / 256), 256>>>(B, N, Bp);
  CK(hipDeviceSynchronize());
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float tot = 0;
    for (int rep = 0; rep < 22; ++rep) {
      CK(hipEventRecord(e0));
      k_gemm_base<<<dim3(N / 128, M / 128), 256>>>(Ap, Bp, Cref);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2) tot += ms;
    }
    printf("%-28s mean %.1f us = %.1f TFLOP/s fp32-equivalent\n", "128x128 (2 WG/CU) baseline", tot / 20 * 1e3, 2.0 * M * N * K / (tot / 20 * 1e-3) / 1e12);
  }
  run<256, 256, 128, 128>("256x256, waves of 128x128", Ap, Bp, C, Cref);
  run<128, 128, 64, 64, 2>("128x128 interleaved, 2 WG/CU", Ap, Bp, C, Cref);
  std::vector<float> hC((size_t)256 * N);
  CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
  double maxabs = 0, scale = 0;
  for (int r = 0; r < 256; r += 17)
    for (int c = 0; c < N; c += 97) {
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * (double)hB[(size_t)c * K + k];
      maxabs = fmax(maxabs, fabs((double)hC[(size_t)r * N + c] - ref)); scale = fmax(scale, fabs(ref));
    }
  printf("max |err| %.3g against float64, largest |score| %.3g -> %.3g of it\n", maxabs, scale, maxabs / scale);
  return 0;
}
