// Feasibility probe (not part of the product): the forward of the BPR step cut along D into 8 slices of D/8 floats,
// one slice per XCD, so that an XCD's L2 holds its 2.7 MB slice of the item table and every gathered byte is an L2 hit.
//   A  = the shape of k_fwd_ugrad (one wave per positive, whole rows, 8 rows in flight)
//   S  = wave per (positive, slice): the 101 item-row slices of the positive sit in 104 VGPRs (four rows per
//        wave-instruction: 16 lanes x 16 B each), partial dots by 16-lane DPP sums, the 8 partial score vectors of a
//        positive are exchanged between the 8 XCDs as 8-byte {tag, value} granules (relaxed agent-scope stores /
//        loads, no fence), every slice-wave then forms the same d loss/d score and its slice of the user-row gradient
//        from the rows it still holds.
//   S1 = S without the exchange (own partials only): the gather floor of this formulation.
//   hipcc -O3 --offload-arch=gfx950 slice_fwd.hip -o slice_fwd && ./slice_fwd
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int D = 512, I = 10677, B = 4096, K = 101, KP = 104, NS = 8, SPIN_MAX = 1 << 16;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {   // every lane of a 16-lane row gets the row's sum
  v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v);
  return v;
}
__device__ __forceinline__ float wsum(float v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ __forceinline__ float dot4(f4 a, f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float softplus(float z) { return fmaxf(z, 0.f) + log1pf(expf(-fabsf(z))); }

__global__ __launch_bounds__(256) void kA(const f4* __restrict__ Iw, const f4* __restrict__ ucur, const float* __restrict__ Ib,
                                          const int* __restrict__ it, float inv_norm, float* __restrict__ g_s,
                                          f4* __restrict__ dU, double* __restrict__ loss_b) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const f4* ur = ucur + (long long)b * (D / 4);
  const f4 u0 = ur[lane], u1 = ur[64 + lane];
  f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  const int* row = it + b * K;
  const f4* p0 = Iw + (long long)row[0] * (D / 4);
  const f4 q0 = p0[lane], q1 = p0[64 + lane];
  const float s0 = wsum(dot4(u0, q0) + dot4(u1, q1)) + Ib[row[0]];
  float gsum = 0.f; double ls = 0.0;
  for (int k0 = 1; k0 < K; k0 += 8) {
    f4 r0[8], r1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) if (k0 + j < K) { const f4* p = Iw + (long long)row[k0 + j] * (D / 4); r0[j] = p[lane]; r1[j] = p[64 + lane]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (k0 + j < K) {
      const float s = wsum(dot4(u0, r0[j]) + dot4(u1, r1[j])) + Ib[row[k0 + j]];
      const float x = s0 - s;
      const float g = inv_norm / (1.f + expf(x));
      a0 += g * r0[j]; a1 += g * r1[j]; gsum += g;
      if (lane == 0) { g_s[b * K + k0 + j] = g; ls += (double)softplus(-x); }
    }
  }
  a0 += -gsum * q0; a1 += -gsum * q1;
  if (lane == 0) { g_s[b * K] = -gsum; loss_b[b] = ls; }
  dU[(long long)b * (D / 4) + lane] = a0; dU[(long long)b * (D / 4) + 64 + lane] = a1;
}

// MODE 0: full | 1: no exchange (own partial x 8)
template <int MODE>
__global__ __launch_bounds__(256) void kS(const f4* __restrict__ Iw, const f4* __restrict__ ucur, const float* __restrict__ Ib,
                                          const int* __restrict__ it, u64* __restrict__ xch, unsigned epoch, float inv_norm,
                                          float* __restrict__ g_s, f4* __restrict__ dU, double* __restrict__ loss_b,
                                          unsigned* __restrict__ tmo) {
  __shared__ float sh[4][KP];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int x = blockIdx.x & 7;                 // slice == XCD under round-robin dispatch
  const int b = (blockIdx.x >> 3) * 4 + w;
  const int sub = lane & 15, grp = lane >> 4;
  const int* row = it + b * K;
  const int id0 = row[lane];
  const int id1 = row[min(64 + lane, K - 1)];
  const f4 uq = ucur[(long long)b * (D / 4) + x * 16 + sub];
  f4 r[26];
#pragma unroll
  for (int j = 0; j < 26; ++j) {
    const int k = min(j * 4 + grp, K - 1);
    const int idv = (j < 16) ? __shfl(id0, k, 64) : __shfl(id1, k - 64, 64);
    r[j] = *(const f4*)((const char*)Iw + (size_t)((unsigned)idv * (unsigned)(D * 4) + (unsigned)(x * 256 + sub * 16)));
  }
  const float b0 = Ib[id0], b1 = Ib[id1];
#pragma unroll
  for (int j = 0; j < 26; ++j) {
    const float p = row16_sum(dot4(uq, r[j]));
    const int k = j * 4 + grp;
    if (sub == 0 && k < K) sh[w][k] = p;
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float p0 = sh[w][lane];
  const float p1 = (64 + lane < K) ? sh[w][64 + lane] : 0.f;
  float s_0 = 0.f, s_1 = 0.f;
  if (MODE == 0 || MODE == 2) {
    u64* mine = xch + ((long long)b * NS + x) * KP;
    __hip_atomic_store(mine + lane, ((u64)epoch << 32) | __float_as_uint(p0), RLX_AGENT);
    if (64 + lane < K) __hip_atomic_store(mine + 64 + lane, ((u64)epoch << 32) | __float_as_uint(p1), RLX_AGENT);
    const u64* all = xch + (long long)b * NS * KP;
    const bool hi = 64 + lane < K;
    float v0[NS], v1[NS];
    for (int spins = 0;; ++spins) {
      bool ok = true;
#pragma unroll
      for (int y = 0; y < NS; ++y) {
        const u64 g = __hip_atomic_load(all + y * KP + lane, RLX_AGENT);
        v0[y] = __uint_as_float((unsigned)g);
        ok &= (unsigned)(g >> 32) == epoch;
        if (hi) {
          const u64 h = __hip_atomic_load(all + y * KP + 64 + lane, RLX_AGENT);
          v1[y] = __uint_as_float((unsigned)h);
          ok &= (unsigned)(h >> 32) == epoch;
        } else v1[y] = 0.f;
      }
      if (__all(ok) || MODE == 2) break;
      if (spins > SPIN_MAX) { if (lane == 0) atomicAdd(tmo, 1u); break; }
      __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int y = 0; y < NS; ++y) { s_0 += v0[y]; s_1 += v1[y]; }
  } else {
    s_0 = p0 * 8.f; s_1 = p1 * 8.f;
  }
  s_0 += b0; s_1 += b1;
  const float s0 = __shfl(s_0, 0, 64);
  const float x0 = s0 - s_0, x1 = s0 - s_1;
  float g_0 = (lane >= 1) ? inv_norm / (1.f + expf(x0)) : 0.f;
  const float g_1 = (64 + lane < K) ? inv_norm / (1.f + expf(x1)) : 0.f;
  const float gsum = wsum(g_0 + g_1);
  if (lane == 0) g_0 = -gsum;
  __builtin_amdgcn_wave_barrier();
  sh[w][lane] = g_0;
  if (lane < KP - 64) sh[w][64 + lane] = g_1;   // zeros behind K
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  f4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 26; ++j) acc += sh[w][j * 4 + grp] * r[j];
#pragma unroll
  for (int q = 0; q < 4; ++q) { acc[q] += __shfl_xor(acc[q], 16, 64); acc[q] += __shfl_xor(acc[q], 32, 64); }
  if (grp == 0) dU[(long long)b * (D / 4) + x * 16 + sub] = acc;
  if (x == (b & 7)) {
    g_s[b * K + lane] = g_0;
    if (64 + lane < K) g_s[b * K + 64 + lane] = g_1;
    double ls = (lane >= 1) ? (double)softplus(-x0) : 0.0;
    if (64 + lane < K) ls += (double)softplus(-x1);
    for (int o = 32; o; o >>= 1) ls += __shfl_xor(ls, o, 64);
    if (lane == 0) loss_b[b] = ls;
  }
}


// S2: two hops.  The slice-wave with x == b % 8 reduces the 8 partial vectors of positive b and publishes d loss/d score;
// the other seven wait for that.  A waiter first polls ONE granule per producer (one 8-lane load), then sweeps.
__device__ __forceinline__ bool probe_wait(const u64* p, int n_lanes, int lane, unsigned epoch, unsigned* tmo) {
  for (int spins = 0;; ++spins) {
    bool ok = true;
    if (lane < n_lanes) ok = (unsigned)(__hip_atomic_load(p, RLX_AGENT) >> 32) == epoch;
    if (__all(ok)) return true;
    if (spins > SPIN_MAX) { if (lane == 0) atomicAdd(tmo, 1u); return false; }
    __builtin_amdgcn_s_sleep(8);
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void kS2(const f4* __restrict__ Iw, const f4* __restrict__ ucur, const float* __restrict__ Ib,
                                           const int* __restrict__ it, u64* __restrict__ xch, u64* __restrict__ gch, unsigned epoch,
                                           float inv_norm, float* __restrict__ g_s, f4* __restrict__ dU,
                                           double* __restrict__ loss_b, unsigned* __restrict__ tmo) {
  __shared__ float sh[4][KP];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int x = blockIdx.x & 7;
  const int b = (blockIdx.x >> 3) * 4 + w;
  const int sub = lane & 15, grp = lane >> 4;
  const bool hi = 64 + lane < K;
  const int* row = it + b * K;
  const int id0 = row[lane];
  const int id1 = row[min(64 + lane, K - 1)];
  const f4 uq = ucur[(long long)b * (D / 4) + x * 16 + sub];
  f4 r[26];
#pragma unroll
  for (int j = 0; j < 26; ++j) {
    const int k = min(j * 4 + grp, K - 1);
    const int idv = (j < 16) ? __shfl(id0, k, 64) : __shfl(id1, k - 64, 64);
    r[j] = *(const f4*)((const char*)Iw + (size_t)((unsigned)idv * (unsigned)(D * 4) + (unsigned)(x * 256 + sub * 16)));
  }
#pragma unroll
  for (int j = 0; j < 26; ++j) {
    const float p = row16_sum(dot4(uq, r[j]));
    const int k = j * 4 + grp;
    if (sub == 0 && k < K) sh[w][k] = p;
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float p0 = sh[w][lane];
  const float p1 = hi ? sh[w][64 + lane] : 0.f;
  const bool reducer = x == (b & 7);
  u64* gout = gch + (long long)b * KP;
  float g_0, g_1;
  if (!reducer) {
    u64* mine = xch + ((long long)b * NS + x) * KP;
    __hip_atomic_store(mine + lane, ((u64)epoch << 32) | __float_as_uint(p0), RLX_AGENT);
    if (hi) __hip_atomic_store(mine + 64 + lane, ((u64)epoch << 32) | __float_as_uint(p1), RLX_AGENT);
    for (int spins = 0;; ++spins) {
      probe_wait(gout + K - 1, 1, lane, epoch, tmo);
      const u64 a = __hip_atomic_load(gout + lane, RLX_AGENT);
      const u64 c = hi ? __hip_atomic_load(gout + 64 + lane, RLX_AGENT) : ((u64)epoch << 32);
      g_0 = __uint_as_float((unsigned)a); g_1 = __uint_as_float((unsigned)c);
      if (__all((unsigned)(a >> 32) == epoch && (unsigned)(c >> 32) == epoch) || spins > 64) break;
    }
  } else {
    const u64* all = xch + (long long)b * NS * KP;
    float s_0 = 0.f, s_1 = 0.f;
    for (int spins = 0;; ++spins) {
      // lane y (< 8, y != x) polls the last granule of producer y
      probe_wait(all + (lane & 7) * KP + K - 1, 8, (lane == x) ? 64 : lane, epoch, tmo);
      bool ok = true;
      s_0 = 0.f; s_1 = 0.f;
      {
        u64 v[NS];
#pragma unroll
        for (int y = 0; y < NS; ++y)
          v[y] = (y == x) ? (((u64)epoch << 32) | __float_as_uint(p0)) : __hip_atomic_load(all + y * KP + lane, RLX_AGENT);
#pragma unroll
        for (int y = 0; y < NS; ++y) { ok &= (unsigned)(v[y] >> 32) == epoch; s_0 += __uint_as_float((unsigned)v[y]); }
      }
      {
        u64 v[NS];
#pragma unroll
        for (int y = 0; y < NS; ++y)
          v[y] = (y == x || !hi) ? (((u64)epoch << 32) | __float_as_uint(p1)) : __hip_atomic_load(all + y * KP + 64 + lane, RLX_AGENT);
#pragma unroll
        for (int y = 0; y < NS; ++y) { ok &= (unsigned)(v[y] >> 32) == epoch; s_1 += __uint_as_float((unsigned)v[y]); }
      }
      if (__all(ok) || spins > 64) break;
    }
    s_0 += Ib[id0]; s_1 += Ib[id1];
    const float s0 = __shfl(s_0, 0, 64);
    const float x0 = s0 - s_0, x1 = s0 - s_1;
    g_0 = (lane >= 1) ? inv_norm / (1.f + expf(x0)) : 0.f;
    g_1 = hi ? inv_norm / (1.f + expf(x1)) : 0.f;
    const float gsum = wsum(g_0 + g_1);
    if (lane == 0) g_0 = -gsum;
    __hip_atomic_store(gout + lane, ((u64)epoch << 32) | __float_as_uint(g_0), RLX_AGENT);
    if (hi) __hip_atomic_store(gout + 64 + lane, ((u64)epoch << 32) | __float_as_uint(g_1), RLX_AGENT);
    g_s[b * K + lane] = g_0;
    if (hi) g_s[b * K + 64 + lane] = g_1;
    double ls = (lane >= 1) ? (double)softplus(-x0) : 0.0;
    if (hi) ls += (double)softplus(-x1);
    for (int o = 32; o; o >>= 1) ls += __shfl_xor(ls, o, 64);
    if (lane == 0) loss_b[b] = ls;
  }
  __builtin_amdgcn_wave_barrier();
  sh[w][lane] = g_0;
  if (lane < KP - 64) sh[w][64 + lane] = hi ? g_1 : 0.f;
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  f4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 26; ++j) acc += sh[w][j * 4 + grp] * r[j];
#pragma unroll
  for (int q = 0; q < 4; ++q) { acc[q] += __shfl_xor(acc[q], 16, 64); acc[q] += __shfl_xor(acc[q], 32, 64); }
  if (grp == 0) dU[(long long)b * (D / 4) + x * 16 + sub] = acc;
}

int main() {
  float *Iw, *ucur, *Ib, *g_s, *dU; int* it; u64* xch; double* loss_b; unsigned* tmo;
  CK(hipMalloc(&Iw, (size_t)I * D * 4)); CK(hipMalloc(&ucur, (size_t)B * D * 4)); CK(hipMalloc(&Ib, I * 4));
  CK(hipMalloc(&it, B * K * 4)); CK(hipMalloc(&g_s, B * K * 4)); CK(hipMalloc(&dU, (size_t)B * D * 4));
  CK(hipMalloc(&xch, (size_t)B * NS * KP * 8)); CK(hipMalloc(&loss_b, B * 8)); CK(hipMalloc(&tmo, 4));
  CK(hipMemset(xch, 0, (size_t)B * NS * KP * 8)); CK(hipMemset(tmo, 0, 4));
  u64* gch; CK(hipMalloc(&gch, (size_t)B * KP * 8)); CK(hipMemset(gch, 0, (size_t)B * KP * 8));
  std::vector<float> hI((size_t)I * D), hU((size_t)B * D), hb(I);
  srand(1);
  for (auto& v : hI) v = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
  for (auto& v : hU) v = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
  for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  CK(hipMemcpy(Iw, hI.data(), hI.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ucur, hU.data(), hU.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(Ib, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  std::vector<int> hi(B * K);
  std::vector<float> refg(B * K), refd((size_t)B * D), gotg(B * K), gotd((size_t)B * D);
  std::vector<double> refl(B), gotl(B);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const float inv_norm = 1.f / (B * (K - 1));
  unsigned epoch = 0;
  const char* nm[5] = {"A per-positive whole rows", "S sliced + exchange", "S1 sliced, no exchange", "S2 sliced, reducer + 2 hops", "S0x one hop, traffic only (no wait, unvalidated)"};
  for (int variant = 0; variant < 5; ++variant) {
    float best = 1e9, tot = 0;
    for (int rep = 0; rep < 12; ++rep) {
      for (auto& v : hi) v = rand() % I;
      CK(hipMemcpy(it, hi.data(), B * K * 4, hipMemcpyHostToDevice));
      CK(hipMemset(dU, 0, (size_t)B * D * 4)); CK(hipMemset(g_s, 0, B * K * 4));
      if ((variant == 1 || variant == 3) && rep == 11) {   // reference for the comparison below
        kA<<<B / 4, 256>>>(( f4*)Iw, (f4*)ucur, Ib, it, inv_norm, g_s, (f4*)dU, loss_b);
        CK(hipMemcpy(refg.data(), g_s, B * K * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(refd.data(), dU, (size_t)B * D * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(refl.data(), loss_b, B * 8, hipMemcpyDeviceToHost));
        CK(hipMemset(dU, 0, (size_t)B * D * 4)); CK(hipMemset(g_s, 0, B * K * 4));
      }
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      if (variant == 0) kA<<<B / 4, 256>>>((f4*)Iw, (f4*)ucur, Ib, it, inv_norm, g_s, (f4*)dU, loss_b);
      else if (variant == 1) kS<0><<<B / 4 * NS, 256>>>((f4*)Iw, (f4*)ucur, Ib, it, xch, ++epoch, inv_norm, g_s, (f4*)dU, loss_b, tmo);
      else if (variant == 2) kS<1><<<B / 4 * NS, 256>>>((f4*)Iw, (f4*)ucur, Ib, it, xch, ++epoch, inv_norm, g_s, (f4*)dU, loss_b, tmo);
      else if (variant == 4) kS<2><<<B / 4 * NS, 256>>>((f4*)Iw, (f4*)ucur, Ib, it, xch, ++epoch, inv_norm, g_s, (f4*)dU, loss_b, tmo);
      else kS2<<<B / 4 * NS, 256>>>((f4*)Iw, (f4*)ucur, Ib, it, xch, gch, ++epoch, inv_norm, g_s, (f4*)dU, loss_b, tmo);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2) { tot += ms; if (ms < best) best = ms; }
    }
    printf("%s: mean %.1f us best %.1f us\n", nm[variant], tot / 10 * 1e3, best * 1e3);
    if (variant == 1 || variant == 3) {
      unsigned ht; CK(hipMemcpy(&ht, tmo, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(gotg.data(), g_s, B * K * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gotd.data(), dU, (size_t)B * D * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(gotl.data(), loss_b, B * 8, hipMemcpyDeviceToHost));
      double eg = 0, mg = 0, ed = 0, md = 0, el = 0;
      for (size_t i = 0; i < refg.size(); ++i) { eg = fmax(eg, fabs(refg[i] - gotg[i])); mg = fmax(mg, fabs(refg[i])); }
      for (size_t i = 0; i < refd.size(); ++i) { ed = fmax(ed, fabs(refd[i] - gotd[i])); md = fmax(md, fabs(refd[i])); }
      for (int i = 0; i < B; ++i) el = fmax(el, fabs(refl[i] - gotl[i]) / fabs(refl[i]));
      printf("  timeouts %u | g_s max err %.3g (max %.3g) | dU max err %.3g (max %.3g) | loss rel err %.3g\n", ht, eg, mg, ed, md, el);
    }
  }
  return 0;
}
