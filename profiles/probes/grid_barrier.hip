// Probe: what does a launch boundary cost against a device-side grid barrier, for a chain of tiny dependent phases
// (the B = 128 training step is two such phases: forward, item/user update)?
//   (a) a replayed hipGraph of 2 * STEPS kernel nodes, each node NB workgroups touching a few rows;
//   (b) ONE kernel looping over the same phases, a grid barrier (device-scope atomic counter + fences) between them.
//   hipcc -O3 --offload-arch=gfx950 grid_barrier.hip -o grid_barrier && ./grid_barrier [NB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int D = 512, ROWS = 8192, STEPS = 64;

__device__ __forceinline__ void phase_body(float* __restrict__ dst, const float* __restrict__ src, int vb, int phase) {
  // each wave: read 4 rows of src (pseudo-random), write one row of dst
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned h = (unsigned)(vb * 4 + wave) * 2654435761u + (unsigned)phase * 40503u;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < 4; ++r) {
    const float* p = src + (size_t)((h >> (r * 3)) % ROWS) * D;
    for (int j = 0; j < 2; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(p + (j * 64 + lane) * 4);
      acc[j * 4 + 0] += v.x; acc[j * 4 + 1] += v.y; acc[j * 4 + 2] += v.z; acc[j * 4 + 3] += v.w;
    }
  }
  float* q = dst + (size_t)(h % ROWS) * D;
  for (int j = 0; j < 2; ++j)
    *reinterpret_cast<float4*>(q + (j * 64 + lane) * 4) = make_float4(acc[j * 4] * 0.25f, acc[j * 4 + 1] * 0.25f, acc[j * 4 + 2] * 0.25f, acc[j * 4 + 3] * 0.25f);
}

__global__ __launch_bounds__(256) void k_phase(float* dst, const float* src, int phase) { phase_body(dst, src, blockIdx.x, phase); }

// every workgroup arrives once per barrier; `gen` counts completed barriers.  Bounded spin: a lost workgroup cannot hang the GPU.
__device__ __forceinline__ bool grid_barrier(unsigned* cnt, unsigned target, int* status) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();   // release: this workgroup's stores leave its XCD's L2
    atomicAdd(cnt, 1u);
    long long spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > 2000000) { ok = false; atomicOr(status, 1); break; }
    }
    __threadfence();   // acquire: drop stale lines
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(256) void k_persistent(float* a, float* b, unsigned* cnt, unsigned base, int* status) {
  const unsigned nb = gridDim.x;
  unsigned done = base;
  for (int s = 0; s < STEPS; ++s) {
    phase_body(b, a, blockIdx.x, 2 * s);
    done += nb;
    if (!grid_barrier(cnt, done, status)) return;
    phase_body(a, b, blockIdx.x, 2 * s + 1);
    done += nb;
    if (!grid_barrier(cnt, done, status)) return;
  }
}

int main(int argc, char** argv) {
  const int NB = argc > 1 ? atoi(argv[1]) : 128;
  float *a, *b; unsigned* cnt; int* status;
  CK(hipMalloc(&a, (size_t)ROWS * D * 4)); CK(hipMalloc(&b, (size_t)ROWS * D * 4));
  CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&status, 4));
  CK(hipMemset(a, 0, (size_t)ROWS * D * 4)); CK(hipMemset(b, 0, (size_t)ROWS * D * 4)); CK(hipMemset(cnt, 0, 4)); CK(hipMemset(status, 0, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int s = 0; s < STEPS; ++s) {
    k_phase<<<NB, 256, 0, st>>>(b, a, 2 * s);
    k_phase<<<NB, 256, 0, st>>>(a, b, 2 * s + 1);
  }
  CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 4; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 2) printf("graph of kernel nodes, %d workgroups: %.2f us per step (2 phases)\n", NB, ms * 1e3 / (4 * STEPS));
  }
  unsigned base = 0;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 4; ++i) { k_persistent<<<NB, 256, 0, st>>>(a, b, cnt, base, status); base += 2u * STEPS * NB; }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int hs; CK(hipMemcpy(&hs, status, 4, hipMemcpyDeviceToHost));
    if (rep >= 2) printf("one kernel with grid barriers, %d workgroups: %.2f us per step (2 phases)  status %d\n", NB, ms * 1e3 / (4 * STEPS), hs);
    if (hs) return 1;
  }
  return 0;
}
