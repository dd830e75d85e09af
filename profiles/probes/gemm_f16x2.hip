// Probe (not part of the product): the 256 x 256 score GEMM core on two fp16 pieces per operand / three products per
// block (csrc/hsk_gemm_wide_h2.h) beside the three-piece bf16 / six-product core (csrc/hsk_gemm_wide.h), both as the
// product's own k loops, with a bare epilogue.  Reports time, fp32-equivalent TFLOP/s and the error of both against
// float64 on a sample of outputs, for gaussian tables at the reference's init scale and for a wide-dynamic-range table.
//   hipcc -O3 --offload-arch=gfx950 -I../../hassaku_amd/csrc -I../../include gemm_f16x2.hip -o gemm_f16x2 && ./gemm_f16x2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "hsk_gemm_wide_h2.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int M = 8192, N = 10752, K = 512;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_split3(const float* __restrict__ X, int rows, __bf16* __restrict__ P) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long r = t / (K / 4);
  const int k = (int)(t - r * (K / 4)) * 4;
  if (r >= rows) return;
  bf16x4 p[3];
  for (int e = 0; e < 4; ++e) {
    const float x = X[r * K + k + e];
    const __bf16 a = (__bf16)x; const float r1 = x - (float)a;
    const __bf16 b = (__bf16)r1; const __bf16 c = (__bf16)(r1 - (float)b);
    p[0][e] = a; p[1][e] = b; p[2][e] = c;
  }
  __bf16* dst = P + ((long long)(k / 16) * rows + r) * 48 + (k % 16);
  for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(dst + 16 * q) = p[q];
}
__global__ void k_absmax(const float* __restrict__ X, long long n, unsigned* out) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float a = fabsf(X[i]);
    if (a > m && a < INFINITY) m = a;
  }
  atomicMax(out, __float_as_uint(m));
}
// X [rows, K] -> [K/16][2][rows][16] fp16 at the scale derived from *amax
__global__ __launch_bounds__(256) void k_split_h2(const float* __restrict__ X, int rows, const unsigned* amax, _Float16* __restrict__ P) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long r = t / (K / 4);
  const int k = (int)(t - r * (K / 4)) * 4;
  if (r >= rows) return;
  const int e = hsk_h2_scale_exp(__uint_as_float(*amax));
  f16x4 hi, lo;
  for (int q = 0; q < 4; ++q) { _Float16 a, b; hsk_split_h2(X[r * K + k + q], e, a, b); hi[q] = a; lo[q] = b; }
  _Float16* dst = P + ((long long)(k / 16) * 2 * rows + r) * 16 + (k % 16);
  *reinterpret_cast<f16x4*>(dst) = hi;
  *reinterpret_cast<f16x4*>(dst + (long long)rows * 16) = lo;
}

__device__ __forceinline__ void store_tile(const hsk_w_f32x16 (&acc)[4][4], float* C, int m0, int n0, int wm, int wn, int r32, int h, float cs, int ldc) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + wm * 128 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h, col = n0 + wn * 128 + j * 32 + r32;
        C[(long long)row * ldc + col] = acc[i][j][q] * cs;
      }
}

template <bool store>
__global__ __launch_bounds__(256, 1) void k_gemm_x3(const __bf16* __restrict__ Apl, const __bf16* __restrict__ Bpl, float* __restrict__ C, int ldc, int kdim) {
  extern __shared__ __attribute__((aligned(16))) __bf16 wlds[];
  __bf16* As = wlds;
  __bf16* Bs = wlds + 2 * GEMM_W_A_STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r32 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 256, n0 = blockIdx.x * 256;
  hsk_w_f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  hsk_wide_stage stg;
  hsk_wide_init(stg, tid);
  const int NT = kdim / 16;
  const __bf16* a0 = Apl + (long long)m0 * 48;
  const __bf16* b0 = Bpl + (long long)n0 * 48;
  const long long a_step = (long long)M * 48, b_step = (long long)N * 48;
  hsk_wide_load(stg, a0, b0, tid);
  hsk_wide_store(stg, As, Bs);
  hsk_wide_load(stg, a0 + a_step, b0 + b_step, tid);
  __syncthreads();
  hsk_wide_kloop(acc, stg, As, Bs, a0, b0, a_step, b_step, NT, tid, wm, wn, r32, h);
  if (store) store_tile(acc, C, m0, n0, wm, wn, r32, h, 1.f, ldc);
  else { float s = 0;
_Pragma("unroll") for (int i = 0; i < 4; ++i)
_Pragma("unroll") for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15]; if (s == 12345.678f) C[0] = s; }
}

template <bool store>
__global__ __launch_bounds__(256, 1) void k_gemm_h2(const _Float16* __restrict__ Apl, const _Float16* __restrict__ Bpl, float* __restrict__ C,
                                                    const unsigned* amax, int a_rows, int b_rows, int ldc, int kdim) {
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
  hsk_h2_kloop(acc, stg, hlds, Apl, Bpl, a_rows, b_rows, m0, n0, NT, tid, wm, wn, r32, h);
  const float cs = ldexpf(1.f, -(hsk_h2_scale_exp(__uint_as_float(amax[0])) + hsk_h2_scale_exp(__uint_as_float(amax[1]))));
  if (store) store_tile(acc, C, m0, n0, wm, wn, r32, h, cs, ldc);
  else { float s = 0;
_Pragma("unroll") for (int i = 0; i < 4; ++i)
_Pragma("unroll") for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15]; if (s == 12345.678f) C[0] = s; }
}

template <typename F>
static float time_us(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float tot = 0;
  for (int rep = 0; rep < 22; ++rep) {
    CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 2) tot += ms;
  }
  return tot / 20 * 1e3f;
}

static void check(const char* what, const float* C, const std::vector<float>& hA, const std::vector<float>& hB) {
  std::vector<float> hC((size_t)M * N);
  CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
  double maxabs = 0, scale = 0, maxrel_terms = 0;
  for (int r = 0; r < M; r += 257)
    for (int c = 0; c < N; c += 97) {
      double ref = 0, sab = 0;
      for (int k = 0; k < K; ++k) { const double p = (double)hA[(size_t)r * K + k] * (double)hB[(size_t)c * K + k]; ref += p; sab += fabs(p); }
      const double err = fabs((double)hC[(size_t)r * N + c] - ref);
      maxabs = fmax(maxabs, err); scale = fmax(scale, fabs(ref)); maxrel_terms = fmax(maxrel_terms, err / sab);
    }
  printf("  %-22s max |err| %.3g = %.3g of the largest |score|; max err / sum|terms| %.3g\n", what, maxabs, maxabs / scale, maxrel_terms);
}

int main() {
  float *A, *B, *C;
  CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
  __bf16 *Ap, *Bp; _Float16 *Ah, *Bh; unsigned* amax;
  CK(hipMalloc(&Ap, (size_t)M * K * 6)); CK(hipMalloc(&Bp, (size_t)N * K * 6));
  CK(hipMalloc(&Ah, (size_t)M * K * 4)); CK(hipMalloc(&Bh, (size_t)N * K * 4)); CK(hipMalloc(&amax, 8));
  CK(hipFuncSetAttribute((const void*)k_gemm_x3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_W_LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)k_gemm_x3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_W_LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)k_gemm_h2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_H_LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)k_gemm_h2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_H_LDS_BYTES));
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  std::mt19937 g(3);
  for (int variant = 0; variant < 2; ++variant) {
    std::normal_distribution<float> nd(0.f, variant == 0 ? 0.1f / K : 0.3f);
    std::lognormal_distribution<float> ln(0.f, 1.5f);
    std::vector<float> colscale(K, 1.f);
    if (variant == 1) for (auto& v : colscale) v = ln(g);
    for (size_t i = 0; i < hA.size(); ++i) hA[i] = nd(g) * colscale[i % K];
    for (size_t i = 0; i < hB.size(); ++i) hB[i] = nd(g) * colscale[i % K];
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    printf("%s\n", variant == 0 ? "gaussian tables, std 0.1 / D (the reference's init)" : "gaussian x lognormal(1.5) column scales (wide dynamic range)");
    const unsigned nb_a = (unsigned)(((size_t)M * K / 4 + 255) / 256), nb_b = (unsigned)(((size_t)N * K / 4 + 255) / 256);
    k_split3<<<nb_a, 256>>>(A, M, Ap);
    k_split3<<<nb_b, 256>>>(B, N, Bp);
    CK(hipMemset(amax, 0, 8));
    k_absmax<<<1024, 256>>>(A, (long long)M * K, amax);
    k_absmax<<<1024, 256>>>(B, (long long)N * K, amax + 1);
    k_split_h2<<<nb_a, 256>>>(A, M, amax, Ah);
    k_split_h2<<<nb_b, 256>>>(B, N, amax + 1, Bh);
    CK(hipDeviceSynchronize());
    const dim3 grid(N / 256, M / 256);
    for (int store = 1; store >= 0; --store) {
      const float t3 = time_us([&] { if (store) k_gemm_x3<true><<<grid, 256, GEMM_W_LDS_BYTES>>>(Ap, Bp, C, N, K); else k_gemm_x3<false><<<grid, 256, GEMM_W_LDS_BYTES>>>(Ap, Bp, C, N, K); });
      if (store) check("bf16 x 3, six products", C, hA, hB);
      const float t2 = time_us([&] { if (store) k_gemm_h2<true><<<grid, 256, GEMM_H_LDS_BYTES>>>(Ah, Bh, C, amax, M, N, N, K); else k_gemm_h2<false><<<grid, 256, GEMM_H_LDS_BYTES>>>(Ah, Bh, C, amax, M, N, N, K); });
      if (store) check("fp16 x 2, three products", C, hA, hB);
      printf("  %s: bf16x3 %.1f us = %.0f TF fp32-equivalent (%.0f TF of bf16 MFMA); fp16x2 %.1f us = %.0f TF fp32-equivalent (%.0f TF of f16 MFMA)\n",
             store ? "with the bare store" : "k loop alone      ", t3, 2.0 * M * N * K / t3 / 1e6, 12.0 * M * N * K / t3 / 1e6, t2,
             2.0 * M * N * K / t2 / 1e6, 6.0 * M * N * K / t2 / 1e6);
    }
  }
  return 0;
}
