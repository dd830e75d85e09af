// feasibility probe: per-positive gather (A) vs item-partitioned-by-XCD gather (B); not part of the product
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int D = 512, I = 10677, U = 69878, B = 4096, K = 101;
__device__ __forceinline__ float wsum(float v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64); return v; }

// A: wave per positive, 101 rows, 8 in flight
__global__ __launch_bounds__(256) void kA(const float4* __restrict__ Iw, const float4* __restrict__ Uw, const int* __restrict__ u32,
                                          const int* __restrict__ it, float4* __restrict__ out) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float4* ur = Uw + (long long)u32[b] * (D / 4);
  float4 u0 = ur[lane], u1 = ur[64 + lane];
  float4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  const int* row = it + b * K;
  for (int k0 = 0; k0 < K; k0 += 8) {
    float4 r0[8], r1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) if (k0 + j < K) { const float4* p = Iw + (long long)row[k0 + j] * (D / 4); r0[j] = p[lane]; r1[j] = p[64 + lane]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (k0 + j < K) {
      float d = u0.x * r0[j].x + u0.y * r0[j].y + u0.z * r0[j].z + u0.w * r0[j].w + u1.x * r1[j].x + u1.y * r1[j].y + u1.z * r1[j].z + u1.w * r1[j].w;
      float g = 1.f / (1.f + __expf(wsum(d)));
      a0.x += g * r0[j].x; a0.y += g * r0[j].y; a0.z += g * r0[j].z; a0.w += g * r0[j].w;
      a1.x += g * r1[j].x; a1.y += g * r1[j].y; a1.z += g * r1[j].z; a1.w += g * r1[j].w;
    }
  }
  out[(long long)b * (D / 4) + lane] = a0; out[(long long)b * (D / 4) + 64 + lane] = a1;
}

// B: wave per (positive, partition x = blockIdx % 8): scans the 101 ids, keeps id % 8 == x, gathers those + u row + row[0]
template <int MODE>
__global__ __launch_bounds__(256) void kB(const float4* __restrict__ Iw, const float4* __restrict__ Uw, const int* __restrict__ u32,
                                          const int* __restrict__ it, float4* __restrict__ out) {
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
  const float4* ur = Uw + (long long)u32[b] * (D / 4);
  const float4* p0 = Iw + (long long)row[0] * (D / 4);
  float4 u0, u1, q0, q1;
  if (MODE == 2) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 t;
    t = __builtin_nontemporal_load((const f4*)(ur + lane)); u0 = make_float4(t.x, t.y, t.z, t.w);
    t = __builtin_nontemporal_load((const f4*)(ur + 64 + lane)); u1 = make_float4(t.x, t.y, t.z, t.w);
    t = __builtin_nontemporal_load((const f4*)(p0 + lane)); q0 = make_float4(t.x, t.y, t.z, t.w);
    t = __builtin_nontemporal_load((const f4*)(p0 + 64 + lane)); q1 = make_float4(t.x, t.y, t.z, t.w);
  } else { u0 = ur[lane]; u1 = ur[64 + lane]; q0 = p0[lane]; q1 = p0[64 + lane]; }
  float s0 = wsum(u0.x * q0.x + u0.y * q0.y + u0.z * q0.z + u0.w * q0.w + u1.x * q1.x + u1.y * q1.y + u1.z * q1.z + u1.w * q1.w);
  float4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  for (int k0 = 0; k0 < n; k0 += 8) {
    float4 r0[8], r1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) if (k0 + j < n) { const float4* p = Iw + (long long)lst[w][k0 + j] * (D / 4); r0[j] = p[lane]; r1[j] = p[64 + lane]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (k0 + j < n) {
      float d = u0.x * r0[j].x + u0.y * r0[j].y + u0.z * r0[j].z + u0.w * r0[j].w + u1.x * r1[j].x + u1.y * r1[j].y + u1.z * r1[j].z + u1.w * r1[j].w;
      float g = 1.f / (1.f + __expf(s0 - wsum(d)));
      a0.x += g * r0[j].x; a0.y += g * r0[j].y; a0.z += g * r0[j].z; a0.w += g * r0[j].w;
      a1.x += g * r1[j].x; a1.y += g * r1[j].y; a1.z += g * r1[j].z; a1.w += g * r1[j].w;
    }
  }
  float4* o = out + ((long long)x * B + b) * (D / 4);
  if (MODE != 1 || s0 == 12345.f) { if (MODE == 2) { typedef float f4 __attribute__((ext_vector_type(4))); f4 t0 = {a0.x, a0.y, a0.z, a0.w}, t1 = {a1.x, a1.y, a1.z, a1.w}; __builtin_nontemporal_store(t0, (f4*)(o + lane)); __builtin_nontemporal_store(t1, (f4*)(o + 64 + lane)); } else { o[lane] = a0; o[64 + lane] = a1; } }
}

int main() {
  float *Iw, *Uw; int *u32, *it; float* out;
  CK(hipMalloc(&Iw, (size_t)I * D * 4)); CK(hipMalloc(&Uw, (size_t)U * D * 4));
  CK(hipMalloc(&u32, B * 4)); CK(hipMalloc(&it, B * K * 4)); CK(hipMalloc(&out, (size_t)8 * B * D * 4));
  CK(hipMemset(Iw, 0, (size_t)I * D * 4)); CK(hipMemset(Uw, 0, (size_t)U * D * 4));
  std::vector<int> hu(B), hi(B * K);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int variant = 0; variant < 5; ++variant) {
    float best = 1e9, tot = 0;
    for (int it_ = 0; it_ < 12; ++it_) {
      for (auto& v : hu) v = rand() % U;
      for (auto& v : hi) v = rand() % I;
      CK(hipMemcpy(u32, hu.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(it, hi.data(), B * K * 4, hipMemcpyHostToDevice));
      CK(hipEventRecord(e0));
      if (variant == 0) kA<<<B / 4, 256>>>((float4*)Iw, (float4*)Uw, u32, it, (float4*)out);
      else if (variant == 1) kB<0><<<B / 4 * 8, 256>>>((float4*)Iw, (float4*)Uw, u32, it, (float4*)out);
      else if (variant == 2) kB<1><<<B / 4 * 8, 256>>>((float4*)Iw, (float4*)Uw, u32, it, (float4*)out);
      else if (variant == 3) kB<2><<<B / 4 * 8, 256>>>((float4*)Iw, (float4*)Uw, u32, it, (float4*)out);
      else kB<3><<<B / 4 * 8, 256>>>((float4*)Iw, (float4*)Uw, u32, it, (float4*)out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it_ >= 2) { tot += ms; if (ms < best) best = ms; }
    }
    const char* nm[5] = {"A per-positive", "B partitioned", "B no-write", "B nt u/p0/out", "B misaligned"}; printf("%s: mean %.1f us best %.1f us\n", nm[variant], tot / 10 * 1e3, best * 1e3);
  }
  return 0;
}
