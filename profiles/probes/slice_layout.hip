// Probe (not part of the product): does the item pass's D-sliced gather suffer from its address pattern?
// k_item_user's workgroups on one XCD read the SAME 1 KB half of every 2 KB user row (rows [B][512] floats, slice s =
// floats [256 s, 256 s + 256)): address bit 10 is constant for everything an XCD touches.  If the L2 spreads lines over
// its channels by low address bits, half the channels would serve the whole gather.  Test: the same gather (4096 rows,
// ~100 random rows per "item", one wave per (item, slice), 8 loads in flight, blockIdx % 8 -> slice-owning XCD) from
//   (A) row-major  [B][512]      : slice s at row * 512 + 256 s     (the product's ucur layout)
//   (B) slice-major [2][B][256]   : slice s at (s * B + row) * 256   (each XCD's 4 MB dense)
//   hipcc -O3 --offload-arch=gfx950 slice_layout.hip -o slice_layout && ./slice_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int D = 512, B = 4096, I = 10677, PER = 39;   // entries per item (ml10m: 413 696 entries over 10 677 items)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int LAYOUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_gather(const float* __restrict__ U, const int* __restrict__ rows, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bid = blockIdx.x, xcd = bid & 7, r = bid >> 3;
  const int slice = xcd & 1, group = r * 4 + (xcd >> 1);
  const int item = group * 4 + wave;
  if (item >= I) return;
  const int* lst = rows + (long long)item * PER;
  const int myrow = lane < PER ? lst[lane] : 0;
  f4 acc = {0, 0, 0, 0};
  for (int j = 0; j < PER; j += 8) {
    f4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = __builtin_amdgcn_readlane(myrow, (j + q) < 63 ? (j + q) : 63);
      const long long off = LAYOUT == 0 ? (long long)row * D + slice * 256 + lane * 4
                                        : ((long long)slice * B + row) * 256 + lane * 4;
      v[q] = (j + q < PER) ? *reinterpret_cast<const f4*>(U + off) : f4{0, 0, 0, 0};
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) acc += v[q];
  }
  *reinterpret_cast<f4*>(out + ((long long)item * 2 + slice) * 256 + lane * 4) = acc;
}

int main() {
  float *U, *out; int* rows;
  CK(hipMalloc(&U, (size_t)B * D * 4)); CK(hipMalloc(&out, (size_t)I * D * 4)); CK(hipMalloc(&rows, (size_t)I * PER * 4));
  std::vector<int> h((size_t)I * PER);
  srand(1);
  for (auto& x : h) x = rand() % B;
  CK(hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(U, 0, (size_t)B * D * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = ((I + 3) / 4 + 3) / 4 * 4 * 2 * 1;   // groups * 2 slices, multiple of 8
  const unsigned nblk = (unsigned)(((I + 3) / 4 + 3) / 4) * 8;
  (void)grid;
  for (int layout = 0; layout < 2; ++layout) {
    float tot = 0;
    for (int rep = 0; rep < 22; ++rep) {
      CK(hipEventRecord(e0));
      if (layout == 0) k_gather<0><<<nblk, 256>>>(U, rows, out); else k_gather<1><<<nblk, 256>>>(U, rows, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2) tot += ms;
    }
    const double bytes = (double)I * PER * 2 * 1024;
    printf("%s: %.1f us = %.2f TB/s\n", layout == 0 ? "row-major  [B][512], an XCD reads one half of every row" : "slice-major [2][B][256], an XCD reads a dense 4 MB      ",
           tot / 20 * 1e3, bytes / (tot / 20 * 1e-3) / 1e12);
  }
  return 0;
}
