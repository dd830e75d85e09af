// Feasibility probe (not part of the product): forward of the BPR step with the item table cut into P range partitions,
// partition q served by the 8/P XCDs {q*8/P ..}: an XCD's L2 then only ever sees I/P item rows (21.9 MB / P).
// No exchange inside the launch: every (positive, partition) unit loads the user row and the positive's item row itself
// (s0 is recomputed by each unit), weighs its own negatives, and writes a PARTIAL user-row gradient and a partial sum of the
// negatives' weights; the user update adds the P partial rows.  The negatives of a positive arrive bucketed by partition
// (their order inside a positive is arbitrary: they are i.i.d.), off[b][q] = first negative of partition q.
//   P = 1 is the shape of k_fwd_ugrad.  Loads in flight per wave: R rows (2 x 16 B per lane each), double-buffered.
//   hipcc -O3 --offload-arch=gfx950 part_fwd.hip -o part_fwd && ./part_fwd
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int D = 512, B = 4096, K = 101;
static int I = 10677;   // argv[1]: item rows (2000000 = the hbm workload: nothing cached, table values left zero)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int CTRL, int RM = 0xF>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, RM, 0xF, false));
}
__device__ __forceinline__ float wsum(float v) {
  v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v);
  v = dpp_add<0x142, 0xA>(v); v = dpp_add<0x143, 0xC>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float dot8(f4 a0, f4 a1, f4 b0, f4 b1) {
  return a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w + a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w;
}
__device__ __forceinline__ float softplus(float z) { return fmaxf(z, 0.f) + log1pf(expf(-fabsf(z))); }

template <int P, int R>
__global__ __launch_bounds__(256) void kP(const f4* __restrict__ Iw, const f4* __restrict__ ucur, const float* __restrict__ Ib,
                                          const int* __restrict__ it, const unsigned char* __restrict__ off, float inv_norm,
                                          float* __restrict__ g_s, f4* __restrict__ dUp, float* __restrict__ gsum_p,
                                          double* __restrict__ loss_p) {
  constexpr int M = 8 / P;   // XCDs per partition
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7, q = xcd / M, s = xcd % M, t = blockIdx.x >> 3;
  const int b = __builtin_amdgcn_readfirstlane((t * M + s) * 4 + w);
  if (b >= B) return;
  const int* row = it + b * K;
  const int beg = (P == 1) ? 1 : __builtin_amdgcn_readfirstlane((int)off[b * (P + 1) + q]);
  const int end = (P == 1) ? K : __builtin_amdgcn_readfirstlane((int)off[b * (P + 1) + q + 1]);
  const f4* ur = ucur + (long long)b * (D / 4);
  const int i0 = __builtin_amdgcn_readfirstlane(row[0]);
  const f4* p0 = Iw + (long long)i0 * (D / 4);
  const f4 u0 = ur[lane], u1 = ur[64 + lane];
  const f4 q0 = p0[lane], q1 = p0[64 + lane];
  const int n = end - beg;
  const int myid = (lane < n) ? row[beg + lane] : i0;     // n <= 64 (P == 1: two rounds below)
  f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  float gsum = 0.f; double ls = 0.0;
  const float s0 = wsum(dot8(u0, u1, q0, q1)) + Ib[i0];
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int nr = min(64, n - c0);
    const int id = (c0 == 0) ? myid : ((lane < nr) ? row[beg + c0 + lane] : i0);
    const float bias = Ib[id];
    float gv = 0.f, xv = 0.f;
    f4 A0[R], A1[R], B0[R], B1[R];
    auto pre = [&](f4(&r0)[R], f4(&r1)[R], int j) {
#pragma unroll
      for (int r = 0; r < R; ++r) if (j + r < nr) {
        const f4* p = Iw + (long long)__builtin_amdgcn_readlane(id, j + r) * (D / 4);
        r0[r] = p[lane]; r1[r] = p[64 + lane];
      }
    };
    auto proc = [&](f4(&r0)[R], f4(&r1)[R], int j) {
#pragma unroll
      for (int r = 0; r < R; ++r) if (j + r < nr) {
        const float sc = wsum(dot8(u0, u1, r0[r], r1[r])) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bias), j + r));
        const float x = s0 - sc;
        const float g = inv_norm / (1.f + expf(x));
        a0 += g * r0[r]; a1 += g * r1[r]; gsum += g;
        gv = (lane == j + r) ? g : gv; xv = (lane == j + r) ? x : xv;
      }
    };
    pre(A0, A1, 0);
    for (int j = 0; j < nr; j += 2 * R) { pre(B0, B1, j + R); proc(A0, A1, j); pre(A0, A1, j + 2 * R); proc(B0, B1, j + R); }
    if (lane < nr) { g_s[b * K + beg + c0 + lane] = gv; ls += (double)softplus(-xv); }
  }
  a0 += -gsum * q0; a1 += -gsum * q1;   // this unit's share of the positive's term: g0 = -sum over ALL partitions
  f4* o = dUp + ((long long)q * B + b) * (D / 4);
  o[lane] = a0; o[64 + lane] = a1;
  for (int o2 = 32; o2; o2 >>= 1) ls += __shfl_xor(ls, o2, 64);
  if (lane == 0) { gsum_p[q * B + b] = gsum; loss_p[q * B + b] = ls; }
}

template <int P, int R>
float run(const float* Iw, const float* ucur, const float* Ib, const int* it, const unsigned char* off, float inv_norm, float* g_s,
          float* dUp, float* gsum_p, double* loss_p, hipEvent_t e0, hipEvent_t e1) {
  CK(hipEventRecord(e0));
  kP<P, R><<<8 * (B / 4) / (8 / P), 256>>>((const f4*)Iw, (const f4*)ucur, Ib, it, off, inv_norm, g_s, (f4*)dUp, gsum_p, loss_p);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f;
}

int main(int argc, char** argv) {
  if (argc > 1) I = atoi(argv[1]);
  const bool big = I > 100000;
  float *Iw, *ucur, *Ib, *g_s, *dUp, *gsum_p; int* it; unsigned char* off; double* loss_p;
  CK(hipMalloc(&Iw, (size_t)I * D * 4)); CK(hipMalloc(&ucur, (size_t)B * D * 4)); CK(hipMalloc(&Ib, I * 4));
  CK(hipMalloc(&it, B * K * 4)); CK(hipMalloc(&g_s, B * K * 4)); CK(hipMalloc(&dUp, (size_t)8 * B * D * 4));
  CK(hipMalloc(&gsum_p, 8 * B * 4)); CK(hipMalloc(&loss_p, 8 * B * 8)); CK(hipMalloc(&off, B * 9));
  std::vector<float> hI(big ? 1 : (size_t)I * D), hU((size_t)B * D), hb(I);
  srand(1);
  for (auto& v : hI) v = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
  for (auto& v : hU) v = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
  for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  if (big) CK(hipMemset(Iw, 0, (size_t)I * D * 4)); else CK(hipMemcpy(Iw, hI.data(), hI.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ucur, hU.data(), hU.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(Ib, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const float inv_norm = 1.f / (B * (K - 1));
  std::vector<int> hi(B * K);
  std::vector<unsigned char> ho(B * 9);
  std::vector<float> ref((size_t)B * D), got((size_t)8 * B * D), refg(B * K), gotg(B * K);
  const int Ps[4] = {1, 2, 4, 8};
  for (int pi = 0; pi < (big ? 1 : 4); ++pi) {
    const int P = Ps[pi];
    for (int Rsel = 0; Rsel < 2; ++Rsel) {
      float tot = 0, best = 1e9;
      for (int rep = 0; rep < 12; ++rep) {
        for (auto& v : hi) v = (int)((((long long)rand() << 15) ^ rand()) % I);
        for (int b = 0; b < B; ++b) {   // bucket the negatives of a positive by partition
          int* r = &hi[b * K];
          std::stable_sort(r + 1, r + K, [&](int a, int c) { return (long long)a * P / I < (long long)c * P / I; });
          int pos = 1;
          for (int q = 0; q < P; ++q) { ho[b * (P + 1) + q] = (unsigned char)pos; while (pos < K && (long long)r[pos] * P / I == q) ++pos; }
          ho[b * (P + 1) + P] = (unsigned char)K;
        }
        CK(hipMemcpy(it, hi.data(), B * K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(off, ho.data(), B * 9, hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        float us = 0;
#define RUN(PP) us = (Rsel == 0) ? run<PP, 4>(Iw, ucur, Ib, it, off, inv_norm, g_s, dUp, gsum_p, loss_p, e0, e1) \
                                 : run<PP, 6>(Iw, ucur, Ib, it, off, inv_norm, g_s, dUp, gsum_p, loss_p, e0, e1)
        if (P == 1) RUN(1); else if (P == 2) RUN(2); else if (P == 4) RUN(4); else RUN(8);
        if (rep >= 2) { tot += us; best = std::min(best, us); }
      }
      printf("P=%d R=%d: mean %.1f us best %.1f us", P, Rsel ? 6 : 4, tot / 10, best);
      // check: sum of the partial rows == the P = 1 result on the same batch (negatives only reordered)
      CK(hipMemcpy(got.data(), dUp, (size_t)P * B * D * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(gotg.data(), g_s, B * K * 4, hipMemcpyDeviceToHost));
      run<1, 4>(Iw, ucur, Ib, it, off, inv_norm, g_s, dUp, gsum_p, loss_p, e0, e1);
      CK(hipMemcpy(ref.data(), dUp, (size_t)B * D * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(refg.data(), g_s, B * K * 4, hipMemcpyDeviceToHost));
      double ed = 0, md = 0, eg = 0;
      for (size_t i = 0; i < (size_t)B * D; ++i) {
        double sum = 0; for (int q = 0; q < P; ++q) sum += got[(size_t)q * B * D + i];
        ed = fmax(ed, fabs(sum - ref[i])); md = fmax(md, fabs(ref[i]));
      }
      for (int b = 0; b < B; ++b) for (int k = 1; k < K; ++k) eg = fmax(eg, fabs(refg[b * K + k] - gotg[b * K + k]));
      printf("   | dU err %.3g (max %.3g), g err %.3g\n", ed, md, eg);
    }
  }
  return 0;
}
