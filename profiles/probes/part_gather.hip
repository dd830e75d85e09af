// Feasibility probe (not part of the product): per-positive row gather (A, the shape of k_fwd_ugrad) against an
// item-partitioned gather (B: wave per (positive, item id mod 8), workgroup index mod 8 == partition, so that under
// round-robin workgroup->XCD dispatch an XCD's L2 only ever sees its eighth of the item table).
//   hipcc -O3 --offload-arch=gfx950 part_gather.hip -o part_gather && ./part_gather
// Measured on MI355X (us, mean of 10): A 102.0 | B 77.9 | B without the partial-row write 73.2 |
// B with nontemporal loads/stores on the once-read rows 101.1 | B with partitions misaligned to the XCDs 104.4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int D = 512, I = 10677, U = 69878, B = 4096, K = 101;
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float wsum(float v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ __forceinline__ float dot8(f4 a0, f4 a1, f4 b0, f4 b1) {
  return a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w + a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w;
}

__global__ __launch_bounds__(256) void kA(const f4* __restrict__ Iw, const f4* __restrict__ Uw, const int* __restrict__ u32,
                                          const int* __restrict__ it, f4* __restrict__ out) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const f4* ur = Uw + (long long)u32[b] * (D / 4);
  const f4 u0 = ur[lane], u1 = ur[64 + lane];
  f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  const int* row = it + b * K;
  for (int k0 = 0; k0 < K; k0 += 8) {
    f4 r0[8], r1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) if (k0 + j < K) { const f4* p = Iw + (long long)row[k0 + j] * (D / 4); r0[j] = p[lane]; r1[j] = p[64 + lane]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (k0 + j < K) {
      const float g = 1.f / (1.f + __expf(wsum(dot8(u0, u1, r0[j], r1[j]))));
      a0 += g * r0[j]; a1 += g * r1[j];
    }
  }
  out[(long long)b * (D / 4) + lane] = a0; out[(long long)b * (D / 4) + 64 + lane] = a1;
}

// MODE 0 partitioned | 1 no partial-row write | 2 nontemporal u / positive row / output | 3 partitions NOT aligned to XCDs
template <int MODE>
__global__ __launch_bounds__(256) void kB(const f4* __restrict__ Iw, const f4* __restrict__ Uw, const int* __restrict__ u32,
                                          const int* __restrict__ it, f4* __restrict__ out) {
  __shared__ int lst[4][128];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int x = (MODE == 3) ? ((blockIdx.x >> 3) & 7) : (blockIdx.x & 7);
  const int b = (MODE == 3) ? (((blockIdx.x >> 6) * 8 + (blockIdx.x & 7)) * 4 + w) : ((blockIdx.x >> 3) * 4 + w);
  if (b >= B) return;
  const int* row = it + b * K;
  int n = 0;
  for (int k0 = 0; k0 < K; k0 += 64) {
    const int k = k0 + lane;
    const int id = (k < K) ? row[k] : -1;
    const bool mine = (k >= 1) && (k < K) && ((id & 7) == x);
    const unsigned long long m = __ballot(mine);
    if (mine) lst[w][n + __popcll(m & ((1ull << lane) - 1ull))] = id;
    n += __popcll(m);
  }
  __builtin_amdgcn_wave_barrier();
  const f4* ur = Uw + (long long)u32[b] * (D / 4);
  const f4* p0 = Iw + (long long)row[0] * (D / 4);
  f4 u0, u1, q0, q1;
  if (MODE == 2) {
    u0 = __builtin_nontemporal_load(ur + lane); u1 = __builtin_nontemporal_load(ur + 64 + lane);
    q0 = __builtin_nontemporal_load(p0 + lane); q1 = __builtin_nontemporal_load(p0 + 64 + lane);
  } else { u0 = ur[lane]; u1 = ur[64 + lane]; q0 = p0[lane]; q1 = p0[64 + lane]; }
  const float s0 = wsum(dot8(u0, u1, q0, q1));
  f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  for (int k0 = 0; k0 < n; k0 += 8) {
    f4 r0[8], r1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) if (k0 + j < n) { const f4* p = Iw + (long long)lst[w][k0 + j] * (D / 4); r0[j] = p[lane]; r1[j] = p[64 + lane]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (k0 + j < n) {
      const float g = 1.f / (1.f + __expf(s0 - wsum(dot8(u0, u1, r0[j], r1[j]))));
      a0 += g * r0[j]; a1 += g * r1[j];
    }
  }
  f4* o = out + ((long long)x * B + b) * (D / 4);
  if (MODE != 1 || s0 == 12345.f) {
    if (MODE == 2) { __builtin_nontemporal_store(a0, o + lane); __builtin_nontemporal_store(a1, o + 64 + lane); }
    else { o[lane] = a0; o[64 + lane] = a1; }
  }
}

int main() {
  float *Iw, *Uw, *out; int *u32, *it;
  CK(hipMalloc(&Iw, (size_t)I * D * 4)); CK(hipMalloc(&Uw, (size_t)U * D * 4));
  CK(hipMalloc(&u32, B * 4)); CK(hipMalloc(&it, B * K * 4)); CK(hipMalloc(&out, (size_t)8 * B * D * 4));
  CK(hipMemset(Iw, 0, (size_t)I * D * 4)); CK(hipMemset(Uw, 0, (size_t)U * D * 4));
  std::vector<int> hu(B), hi(B * K);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* nm[5] = {"A per-positive", "B partitioned", "B no-write", "B nt u/p0/out", "B misaligned"};
  for (int variant = 0; variant < 5; ++variant) {
    float best = 1e9, tot = 0;
    for (int rep = 0; rep < 12; ++rep) {
      for (auto& v : hu) v = rand() % U;
      for (auto& v : hi) v = rand() % I;
      CK(hipMemcpy(u32, hu.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(it, hi.data(), B * K * 4, hipMemcpyHostToDevice));
      CK(hipEventRecord(e0));
      if (variant == 0) kA<<<B / 4, 256>>>((f4*)Iw, (f4*)Uw, u32, it, (f4*)out);
      else if (variant == 1) kB<0><<<B / 4 * 8, 256>>>((f4*)Iw, (f4*)Uw, u32, it, (f4*)out);
      else if (variant == 2) kB<1><<<B / 4 * 8, 256>>>((f4*)Iw, (f4*)Uw, u32, it, (f4*)out);
      else if (variant == 3) kB<2><<<B / 4 * 8, 256>>>((f4*)Iw, (f4*)Uw, u32, it, (f4*)out);
      else kB<3><<<B / 4 * 8, 256>>>((f4*)Iw, (f4*)Uw, u32, it, (f4*)out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2) { tot += ms; if (ms < best) best = ms; }
    }
    printf("%s: mean %.1f us best %.1f us\n", nm[variant], tot / 10 * 1e3, best * 1e3);
  }
  return 0;
}
