// Probe (not part of the product): what the item pass of the ml10m step is made of.  k_item_user's item workgroups take
// 70 us alone for an 847 MB gather of user-row slices that, by itself, runs in 36 us (slice_layout.hip).  The same
// gather here with the pass's other ingredients switched on one by one:
//   CHAIN  the entry list behind two dependent loads (offsets -> perm -> {weight, row index}) instead of a direct list
//   RMW    the item's own row slice of p, m, v loaded before the gather and stored after it (131 MB read + written)
//   ZIPF   list lengths from the batch's law (101 entries per positive: the positive by popularity rank^-0.8, 100 uniform)
//   hipcc -O3 --offload-arch=gfx950 item_pass_parts.hip -o item_pass_parts && ./item_pass_parts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int D = 512, B = 4096, I = 10677, K = 101;
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool CHAIN, bool RMW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_item(const float* __restrict__ U, const int* __restrict__ offsets, const int* __restrict__ perm,
            const float* __restrict__ g_s, const int* __restrict__ direct_rows, const float* __restrict__ direct_w,
            float* __restrict__ P, float* __restrict__ M, float* __restrict__ V) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bid = blockIdx.x, xcd = bid & 7, r = bid >> 3;
  const int slice = xcd & 1, group = r * 4 + (xcd >> 1);
  const int item = group * 4 + wave;
  if (item >= I) return;
  const int beg = offsets[item], end = offsets[item + 1];
  const long long own = (long long)item * D + slice * 256 + lane * 4;
  f4 p = {0, 0, 0, 0}, m = p, v = p;
  if (RMW) {
    p = *reinterpret_cast<const f4*>(P + own);
    m = *reinterpret_cast<const f4*>(M + own);
    v = *reinterpret_cast<const f4*>(V + own);
  }
  f4 acc = {0, 0, 0, 0};
  for (int c0 = beg; c0 < end; c0 += 64) {
    const int nr = min(64, end - c0);
    int myrow = 0;
    float myg = 0.f;
    if (lane < nr) {
      if (CHAIN) {
        const int e = perm[c0 + lane];
        myg = g_s[e];
        myrow = e / K;
      } else {
        myrow = direct_rows[c0 + lane];
        myg = direct_w[c0 + lane];
      }
    }
    for (int j = 0; j < nr; j += 8) {
      f4 val[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int row = __builtin_amdgcn_readlane(myrow, (j + q) < 63 ? (j + q) : 63);
        val[q] = (j + q < nr) ? *reinterpret_cast<const f4*>(U + (long long)row * D + slice * 256 + lane * 4) : f4{0, 0, 0, 0};
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myg), (j + q) < 63 ? (j + q) : 63));
        if (j + q < nr) acc += g * val[q];
      }
    }
  }
  if (RMW) {
    m = 0.9f * m + 0.1f * acc;
    v = 0.999f * v + 0.001f * acc * acc;
    p = p - 1e-3f * m / (__builtin_elementwise_sqrt(v) + 1e-8f);
    *reinterpret_cast<f4*>(P + own) = p;
    *reinterpret_cast<f4*>(M + own) = m;
    *reinterpret_cast<f4*>(V + own) = v;
  } else {
    *reinterpret_cast<f4*>(P + own) = acc;
  }
}

int main() {
  srand(1);
  for (int zipf = 0; zipf < 2; ++zipf) {
    // the batch: B positives x K entries; entry e = b * K + k names item it[e]
    std::vector<int> it((size_t)B * K);
    std::vector<double> cdf(I);
    double tot = 0;
    for (int i = 0; i < I; ++i) { tot += std::pow(i + 1.0, -0.8); cdf[i] = tot; }
    for (int b = 0; b < B; ++b)
      for (int k = 0; k < K; ++k) {
        int item;
        if (zipf && k == 0) {
          const double x = (rand() / (double)RAND_MAX) * tot;
          item = (int)(std::lower_bound(cdf.begin(), cdf.end(), x) - cdf.begin());
          if (item >= I) item = I - 1;
        } else {
          item = rand() % I;
        }
        it[(size_t)b * K + k] = item;
      }
    std::vector<int> offs(I + 1, 0), perm(it.size()), rows(it.size());
    for (int x : it) offs[x + 1]++;
    for (int i = 0; i < I; ++i) offs[i + 1] += offs[i];
    std::vector<int> cur(offs.begin(), offs.end() - 1);
    for (size_t e = 0; e < it.size(); ++e) { perm[cur[it[e]]] = (int)e; rows[cur[it[e]]] = (int)(e / K); cur[it[e]]++; }
    int longest = 0;
    for (int i = 0; i < I; ++i) longest = std::max(longest, offs[i + 1] - offs[i]);
    float *U, *P, *M, *V, *gs, *dw; int *doffs, *dperm, *drows;
    CK(hipMalloc(&U, (size_t)B * D * 4)); CK(hipMalloc(&P, (size_t)I * D * 4)); CK(hipMalloc(&M, (size_t)I * D * 4));
    CK(hipMalloc(&V, (size_t)I * D * 4)); CK(hipMalloc(&gs, it.size() * 4)); CK(hipMalloc(&dw, it.size() * 4));
    CK(hipMalloc(&doffs, (I + 1) * 4)); CK(hipMalloc(&dperm, it.size() * 4)); CK(hipMalloc(&drows, it.size() * 4));
    CK(hipMemset(U, 0, (size_t)B * D * 4)); CK(hipMemset(P, 0, (size_t)I * D * 4)); CK(hipMemset(M, 0, (size_t)I * D * 4));
    CK(hipMemset(V, 0, (size_t)I * D * 4)); CK(hipMemset(gs, 0, it.size() * 4)); CK(hipMemset(dw, 0, it.size() * 4));
    CK(hipMemcpy(doffs, offs.data(), (I + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dperm, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(drows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned nblk = (unsigned)(((I + 3) / 4 + 3) / 4) * 8;
    for (int var = 0; var < 4; ++var) {
      float t = 0;
      for (int rep = 0; rep < 22; ++rep) {
        CK(hipEventRecord(e0));
        switch (var) {
          case 0: k_item<false, false><<<nblk, 256>>>(U, doffs, dperm, gs, drows, dw, P, M, V); break;
          case 1: k_item<true, false><<<nblk, 256>>>(U, doffs, dperm, gs, drows, dw, P, M, V); break;
          case 2: k_item<false, true><<<nblk, 256>>>(U, doffs, dperm, gs, drows, dw, P, M, V); break;
          case 3: k_item<true, true><<<nblk, 256>>>(U, doffs, dperm, gs, drows, dw, P, M, V); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) t += ms;
      }
      printf("%s lists (longest %d)  chain %d  rmw %d : %.1f us\n", zipf ? "batch-law" : "uniform  ", longest, var & 1, var >> 1, t / 20 * 1e3);
    }
    (void)hipFree(U); (void)hipFree(P); (void)hipFree(M); (void)hipFree(V); (void)hipFree(gs); (void)hipFree(dw);
    (void)hipFree(doffs); (void)hipFree(dperm); (void)hipFree(drows);
  }
  return 0;
}
