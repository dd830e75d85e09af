// hsk_synth.hip -- synthetic training interactions generated straight into HBM (BASELINE configs[4]: 100 M users x
// 10 M items, ~20 positives per user; SURVEY 8d: "interactions generated on device, do not materialise CSVs").
//
// Stands where the reference's TrainRecDataset._prepare_data builds its COO / CSR matrices from the CSV files
// (data/dataset.py:120-131): the outputs are exactly the arrays the sampler and the fused step take -- csr_indptr
// int64 [U+1], csr_indices int32 [nnz] (sorted and duplicate-free inside a row), coo_user int32 [nnz]; the COO order IS
// the CSR order, so coo_item aliases csr_indices.  Everything is a pure function of (seed, user id): every rank of a job
// generates the same arrays without a broadcast, and oracle/oracle.py restates the law in numpy for the tests.
//
//   deg(u)    = deg_min + Philox(seed; u, 0xD).x mod deg_span
//   item(u,j) = floor(I * x^skew),  x = (j + r_j) / deg(u),  r_j = (Philox(seed; u, j/4, 0xE)[j%4] + 0.5) / 2^32
//               one draw per stratum [j/deg, (j+1)/deg): sorted by construction; x^skew (skew = 1, 2, 3 by repeated
//               multiplication -- no libm, so host and device agree bit for bit) makes low item ids popular;
//               collisions at stratum borders are resolved by item_j = max(item_j, item_{j-1} + 1), the tail is
//               clamped back below I.
#include "hsk_common.h"

#define HSK_SYNTH_MAX_DEG 64

__host__ __device__ __forceinline__ int hsk_synth_deg(uint64_t u, int deg_min, int deg_span, uint64_t seed) {
  hsk_u32x4 c = {(uint32_t)u, (uint32_t)(u >> 32), 0u, 0xDu};
  const hsk_u32x4 r = hsk_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return deg_min + (int)(r.x % (uint32_t)deg_span);
}

__global__ __launch_bounds__(256) void k_synth_degrees(long long n_users, int deg_min, int deg_span, uint64_t seed,
                                                       int64_t* __restrict__ indptr) {
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u == 0) indptr[0] = 0;
  if (u < n_users) indptr[u + 1] = hsk_synth_deg((uint64_t)u, deg_min, deg_span, seed);
}

__global__ __launch_bounds__(256) void k_synth_fill(long long n_users, long long n_items, int skew, uint64_t seed,
                                                    const int64_t* __restrict__ indptr, int32_t* __restrict__ indices,
                                                    int32_t* __restrict__ coo_user) {
#pragma clang fp contract(off)
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_users) return;
  const long long lo = indptr[u];
  const int deg = (int)min((long long)HSK_SYNTH_MAX_DEG, (long long)(indptr[u + 1] - lo));   // (a foreign indptr cannot overrun `it`)
  int it[HSK_SYNTH_MAX_DEG];
  hsk_u32x4 r = {0, 0, 0, 0};
#pragma unroll 1
  for (int j = 0; j < deg; ++j) {
    if ((j & 3) == 0) {
      hsk_u32x4 c = {(uint32_t)u, (uint32_t)((unsigned long long)u >> 32), (uint32_t)(j >> 2), 0xEu};
      r = hsk_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    }
    const uint32_t w = (j & 3) == 0 ? r.x : (j & 3) == 1 ? r.y : (j & 3) == 2 ? r.z : r.w;
    const double rj = ((double)w + 0.5) * (1.0 / 4294967296.0);
    const double x = ((double)j + rj) / (double)deg;
    double t = x;
    for (int k = 1; k < skew; ++k) t = t * x;
    long long v = (long long)(t * (double)n_items);
    if (v > n_items - 1) v = n_items - 1;
    if (j > 0 && v <= it[j - 1]) v = it[j - 1] + 1;
    it[j] = (int)v;
  }
  for (int j = deg - 1; j >= 0; --j) {   // tail back below n_items (deg <= n_items: checked by the host)
    const int cap = (j == deg - 1) ? (int)(n_items - 1) : it[j + 1] - 1;
    if (it[j] > cap) it[j] = cap;
  }
  for (int j = 0; j < deg; ++j) {
    indices[lo + j] = it[j];
    if (coo_user) coo_user[lo + j] = (int32_t)u;
  }
}

extern "C" int hsk_synth_degrees(int64_t n_users, int32_t deg_min, int32_t deg_span, uint64_t seed, int64_t* indptr,
                                 hsk_stream_t stream_) {
  HSK_REQUIRE(indptr != nullptr, HSK_ERR_INVALID, "indptr is NULL");
  HSK_REQUIRE(n_users > 0 && n_users < 0x7fffffff, HSK_ERR_INVALID, "n_users %lld outside (0, 2^31)", (long long)n_users);
  HSK_REQUIRE(deg_min >= 1 && deg_span >= 1 && deg_min + deg_span - 1 <= HSK_SYNTH_MAX_DEG, HSK_ERR_INVALID,
              "degrees [%d, %d) outside [1, %d]", deg_min, deg_min + deg_span, HSK_SYNTH_MAX_DEG);
  k_synth_degrees<<<(unsigned)hsk_ceil_div(n_users, 256), 256, 0, (hipStream_t)stream_>>>(
      (long long)n_users, deg_min, deg_span, seed, indptr);
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

extern "C" int hsk_synth_fill(int64_t n_users, int64_t n_items, int32_t deg_max, int32_t skew, uint64_t seed,
                              const int64_t* indptr, int32_t* indices, int32_t* coo_user, hsk_stream_t stream_) {
  HSK_REQUIRE(indptr && indices, HSK_ERR_INVALID, "indptr / indices must not be NULL");
  HSK_REQUIRE(n_users > 0 && n_users < 0x7fffffff && n_items > 0 && n_items < 0x7fffffff, HSK_ERR_INVALID,
              "bad n_users / n_items");
  HSK_REQUIRE(deg_max >= 1 && deg_max <= HSK_SYNTH_MAX_DEG && deg_max <= n_items, HSK_ERR_INVALID,
              "deg_max %d outside [1, min(%d, n_items)]", deg_max, HSK_SYNTH_MAX_DEG);
  HSK_REQUIRE(skew >= 1 && skew <= 3, HSK_ERR_INVALID, "skew must be 1, 2 or 3");
  k_synth_fill<<<(unsigned)hsk_ceil_div(n_users, 256), 256, 0, (hipStream_t)stream_>>>(
      (long long)n_users, (long long)n_items, skew, seed, indptr, indices, coo_user);
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}
