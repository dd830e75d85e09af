// Feasibility probe (not part of the product): fp32-accurate scores C = A . B^T from bf16 MFMAs.
// Every fp32 operand is cut into three bf16 pieces (x = x1 + x2 + x3: 8 + 8 + 8 significant bits, the cuts are exact in
// fp32) while its tile is staged into LDS; of the nine products per (a, b) pair the six with weight >= 2^-16 are kept --
// (1,1) (1,2) (2,1) (1,3) (3,1) (2,2) -- each a v_mfma_f32_32x32x16_bf16 into the same fp32 accumulator.  A bf16 x bf16
// product is exact in fp32, the dropped terms are <= 2^-24 of |a||b|: the result has fp32-GEMM accuracy, at six bf16
// MFMAs (32 cycles each for 32x32x16) against sixteen fp32 MFMAs (64 cycles each for 32x32x2) per 32x32x32 block: 2.67x
// the fp32-MFMA peak on paper.
//   hipcc -O3 --offload-arch=gfx950 gemm_bf16x3.hip -o gemm_bf16x3 && ./gemm_bf16x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int M = 8192, N = 10752, K = 512;
constexpr int BM = 128, BN = 128, BK = 32, LDK = BK + 8;   // LDS row stride in bf16 elements (80 B: conflict-free b128 reads)
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

__global__ __launch_bounds__(256, 2) void k_gemm_bf16x3(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C) {
  __shared__ __attribute__((aligned(16))) __bf16 As[3][BM * LDK];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[3][BN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int srow = tid >> 3, scol = (tid & 7) * 4;
  const int r32 = lane & 31, h = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  float4 ra[4], rb[4];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      ra[p] = *reinterpret_cast<const float4*>(A + (long long)(m0 + srow + 32 * p) * K + k0 + scol);
      rb[p] = *reinterpret_cast<const float4*>(B + (long long)(n0 + srow + 32 * p) * K + k0 + scol);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = srow + 32 * p;
      bf16x4 a1, a2, a3, b1, b2, b3;
      const float av[4] = {ra[p].x, ra[p].y, ra[p].z, ra[p].w}, bv[4] = {rb[p].x, rb[p].y, rb[p].z, rb[p].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        __bf16 x1, x2, x3;
        split3(av[e], x1, x2, x3); a1[e] = x1; a2[e] = x2; a3[e] = x3;
        split3(bv[e], x1, x2, x3); b1[e] = x1; b2[e] = x2; b3[e] = x3;
      }
      *reinterpret_cast<bf16x4*>(&As[0][row * LDK + scol]) = a1;
      *reinterpret_cast<bf16x4*>(&As[1][row * LDK + scol]) = a2;
      *reinterpret_cast<bf16x4*>(&As[2][row * LDK + scol]) = a3;
      *reinterpret_cast<bf16x4*>(&Bs[0][row * LDK + scol]) = b1;
      *reinterpret_cast<bf16x4*>(&Bs[1][row * LDK + scol]) = b2;
      *reinterpret_cast<bf16x4*>(&Bs[2][row * LDK + scol]) = b3;
    }
  };
  load_tile(0);
  store_tile();
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool has_next = k0 + BK < K;
    if (has_next) load_tile(k0 + BK);
#pragma unroll
    for (int s = 0; s < 2; ++s) {   // two k16 steps per tile
      bf16x8 a[3][2], b[3][2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[pl][i] = *reinterpret_cast<const bf16x8*>(&As[pl][(wm * 64 + i * 32 + r32) * LDK + 16 * s + 8 * h]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[pl][j] = *reinterpret_cast<const bf16x8*>(&Bs[pl][(wn * 64 + j * 32 + r32) * LDK + 16 * s + 8 * h]);
      }
      // smallest terms first
      constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]][i], b[TB[t]][j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (has_next) {
      store_tile();
      __syncthreads();
    }
  }
  // C/D layout of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        const int col = n0 + wn * 64 + j * 32 + r32;
        C[(long long)row * N + col] = acc[i][j][q];
      }
}


// Variant: the pieces made once by a pre-pass, laid out [k-tile][row][piece][32] (a 128-row tile's three pieces of one
// k-tile = 24 KB of consecutive bytes); the GEMM loop only moves 16-byte chunks global -> registers -> LDS.
__global__ __launch_bounds__(256) void k_presplit(const float* __restrict__ X, int rows, __bf16* __restrict__ P) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int per_row = K / 4;
  const long long r = t / per_row;
  const int k = (int)(t - r * per_row) * 4;
  if (r >= rows) return;
  const float4 x = *reinterpret_cast<const float4*>(X + r * K + k);
  const float v[4] = {x.x, x.y, x.z, x.w};
  bf16x4 p1, p2, p3;
  for (int e = 0; e < 4; ++e) { __bf16 a, b, c; split3(v[e], a, b, c); p1[e] = a; p2[e] = b; p3[e] = c; }
  __bf16* dst = P + ((long long)(k / BK) * rows + r) * 96 + (k % BK);
  *reinterpret_cast<bf16x4*>(dst) = p1;
  *reinterpret_cast<bf16x4*>(dst + 32) = p2;
  *reinterpret_cast<bf16x4*>(dst + 64) = p3;
}

__global__ __launch_bounds__(256, 2) void k_gemm_planes(const __bf16* __restrict__ Ap, const __bf16* __restrict__ Bp, float* __restrict__ C) {
  __shared__ __attribute__((aligned(16))) __bf16 As[3][BM * LDK];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[3][BN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int r32 = lane & 31, h = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  u32x4 ra[6], rb[6];   // (ext-vector type: HIP's uint4 struct array ended up in scratch here)
  auto load_tile = [&](int k0) {
    const __bf16* a = Ap + ((long long)(k0 / BK) * M + m0) * 96;
    const __bf16* b = Bp + ((long long)(k0 / BK) * N + n0) * 96;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      ra[i] = *reinterpret_cast<const u32x4*>(a + (tid + 256 * i) * 8);
      rb[i] = *reinterpret_cast<const u32x4*>(b + (tid + 256 * i) * 8);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int c = tid + 256 * i, r = c / 12, j = c - r * 12, pl = j >> 2, kc = (j & 3) * 8;
      *reinterpret_cast<u32x4*>(&As[pl][r * LDK + kc]) = ra[i];
      *reinterpret_cast<u32x4*>(&Bs[pl][r * LDK + kc]) = rb[i];
    }
  };
  load_tile(0);
  store_tile();
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool has_next = k0 + BK < K;
    if (has_next) load_tile(k0 + BK);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[3][2], b[3][2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[pl][i] = *reinterpret_cast<const bf16x8*>(&As[pl][(wm * 64 + i * 32 + r32) * LDK + 16 * s + 8 * h]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[pl][j] = *reinterpret_cast<const bf16x8*>(&Bs[pl][(wn * 64 + j * 32 + r32) * LDK + 16 * s + 8 * h]);
      }
      constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]][i], b[TB[t]][j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (has_next) {
      store_tile();
      __syncthreads();
    }
  }
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        const int col = n0 + wn * 64 + j * 32 + r32;
        C[(long long)row * N + col] = acc[i][j][q];
      }
}


// Variant 3: pieces laid out [k16-tile][row][piece][16] (a 128-row tile = 12 KB contiguous), LDS double-buffered at
// BK = 16: the stores of tile t+1 go to the other buffer while tile t is multiplied -- ONE barrier per k-step, and no
// phase in which the workgroup only stores.
constexpr int BK2 = 16, LDK2 = BK2 + 8;   // 48-byte rows: b128 reads conflict-free
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
  __bf16* dst = P + ((long long)(k / BK2) * rows + r) * 48 + (k % BK2);
  *reinterpret_cast<bf16x4*>(dst) = p1;
  *reinterpret_cast<bf16x4*>(dst + 16) = p2;
  *reinterpret_cast<bf16x4*>(dst + 32) = p3;
}

__global__ __launch_bounds__(256, 2) void k_gemm_planes2(const __bf16* __restrict__ Ap, const __bf16* __restrict__ Bp, float* __restrict__ C) {
  __shared__ __attribute__((aligned(16))) __bf16 As[2][3][BM * LDK2];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2][3][BN * LDK2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int r32 = lane & 31, h = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  u32x4 ra[3], rb[3];
  // chunk c of the tile's 768: row c / 6, piece (c % 6) / 2, half (c % 6) % 2
  int lds_off[3], lds_pl[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = tid + 256 * i, r = c / 6, j = c - r * 6;
    lds_pl[i] = j >> 1;
    lds_off[i] = r * LDK2 + (j & 1) * 8;
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
  constexpr int NT = K / BK2;
  load_tile(0);
  store_tile(0);
  load_tile(1);
  __syncthreads();
  for (int t = 0; t < NT; ++t) {
    const int buf = t & 1;
    if (t + 1 < NT) store_tile(buf ^ 1);   // tile t+1 (its loads were issued one step ago); buf^1 was last read in step t-1
    if (t + 2 < NT) load_tile(t + 2);
    bf16x8 a[3][2], b[3][2];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[pl][i] = *reinterpret_cast<const bf16x8*>(&As[buf][pl][(wm * 64 + i * 32 + r32) * LDK2 + 8 * h]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[pl][j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][pl][(wn * 64 + j * 32 + r32) * LDK2 + 8 * h]);
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
  float *A, *B, *C;
  CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  srand(3);
  for (auto& v : hA) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  for (auto& v : hB) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  dim3 grid(N / BN, M / BM);
  float best = 1e9, tot = 0;
  for (int rep = 0; rep < 22; ++rep) {
    CK(hipEventRecord(e0));
    k_gemm_bf16x3<<<grid, 256>>>(A, B, C);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 2) { tot += ms; if (ms < best) best = ms; }
  }
  const double flop = 2.0 * M * N * K;
  printf("bf16x3 (6 terms): mean %.1f us best %.1f us = %.1f TFLOP/s fp32-equivalent\n", tot / 20 * 1e3, best * 1e3, flop / (tot / 20 * 1e-3) / 1e12);
  {
    __bf16 *Ap, *Bp; float* C2;
    CK(hipMalloc(&Ap, (size_t)M * K * 6)); CK(hipMalloc(&Bp, (size_t)N * K * 6)); CK(hipMalloc(&C2, (size_t)M * N * 4));
    float bp = 1e9, tp = 0, tg = 0;
    for (int rep = 0; rep < 22; ++rep) {
      hipEvent_t e2; CK(hipEventCreate(&e2));
      CK(hipEventRecord(e0));
      k_presplit<<<(unsigned)(((size_t)M * K / 4 + 255) / 256), 256>>>(A, M, Ap);
      k_presplit<<<(unsigned)(((size_t)N * K / 4 + 255) / 256), 256>>>(B, N, Bp);
      CK(hipEventRecord(e2));
      k_gemm_planes<<<grid, 256>>>(Ap, Bp, C2);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms, msg; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&msg, e2, e1));
      if (rep >= 2) { tp += ms; tg += msg; if (ms < bp) bp = ms; }
    }
    printf("pre-split planes: mean %.1f us with the pre-pass (GEMM alone %.1f us) = %.1f TFLOP/s fp32-equivalent\n", tp / 20 * 1e3, tg / 20 * 1e3, flop / (tp / 20 * 1e-3) / 1e12);
    std::vector<float> c1((size_t)64 * N), c2((size_t)64 * N);
    CK(hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), C2, c2.size() * 4, hipMemcpyDeviceToHost));
    size_t diff = 0; for (size_t i = 0; i < c1.size(); ++i) diff += c1[i] != c2[i];
    printf("planes vs in-loop split: %zu of %zu values differ\n", diff, c1.size());
  }
  {
    __bf16 *Ap, *Bp; float* C2;
    CK(hipMalloc(&Ap, (size_t)M * K * 6)); CK(hipMalloc(&Bp, (size_t)N * K * 6)); CK(hipMalloc(&C2, (size_t)M * N * 4));
    float tp = 0, tg = 0;
    for (int rep = 0; rep < 22; ++rep) {
      hipEvent_t e2; CK(hipEventCreate(&e2));
      CK(hipEventRecord(e0));
      k_presplit16<<<(unsigned)(((size_t)M * K / 4 + 255) / 256), 256>>>(A, M, Ap);
      k_presplit16<<<(unsigned)(((size_t)N * K / 4 + 255) / 256), 256>>>(B, N, Bp);
      CK(hipEventRecord(e2));
      k_gemm_planes2<<<grid, 256>>>(Ap, Bp, C2);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms, msg; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&msg, e2, e1));
      if (rep >= 2) { tp += ms; tg += msg; }
    }
    printf("planes, LDS double-buffered at BK=16: mean %.1f us with the pre-pass (GEMM alone %.1f us) = %.1f TFLOP/s fp32-equivalent\n", tp / 20 * 1e3, tg / 20 * 1e3, flop / (tp / 20 * 1e-3) / 1e12);
    std::vector<float> c1((size_t)64 * N), c2((size_t)64 * N);
    CK(hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), C2, c2.size() * 4, hipMemcpyDeviceToHost));
    size_t diff = 0; for (size_t i = 0; i < c1.size(); ++i) diff += c1[i] != c2[i];
    printf("double-buffered vs in-loop split: %zu of %zu values differ\n", diff, c1.size());
  }
  std::vector<float> hC((size_t)256 * N);
  CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
  double maxrel = 0, maxabs = 0, scale = 0;
  for (int r = 0; r < 256; r += 17)
    for (int c = 0; c < N; c += 97) {
      double ref = 0, f32 = 0;
      for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * (double)hB[(size_t)c * K + k];
      (void)f32;
      const double err = fabs((double)hC[(size_t)r * N + c] - ref);
      maxabs = fmax(maxabs, err); scale = fmax(scale, fabs(ref));
    }
  maxrel = maxabs / scale;
  printf("max |err| %.3g against float64, largest |score| %.3g -> %.3g of it (fp32 epsilon 6e-8)\n", maxabs, scale, maxrel);
  return 0;
}
