// hsk_rows.h -- embedding-row register tiles, dimension dispatch and small device helpers shared by the
// training kernels.  One wavefront owns one row: lane l holds elements (c*64 + l)*V .. +V of chunk c.
#pragma once
#include "hsk_common.h"
#include <type_traits>

#define HSK_OWNER_NONE 0x7fffffff

// Device-resident step descriptor.  A replayed HIP graph has its kernel arguments frozen at capture time, so what
// changes from one replay to the next -- where the run of batches starts, how many steps were applied before it (RNG
// stream id, Adam bias corrections), the epoch permutation -- is read from here; a kernel of step `rel` of the run
// adds its own (frozen) offset.  desc == NULL: the immediate arguments apply (eager launches).
struct hsk_step_desc {
  long long start0;        // position of the run's first batch in `order`
  const int64_t* order;    // epoch permutation (NULL: identity)
  int step0;               // optimiser steps applied before the run
  int pad;
};

// ---------------------------------------------------------------------------------------------
// dimension dispatch: V floats per lane per chunk, NCH chunks per row, FULL = no tail predicate
// ---------------------------------------------------------------------------------------------
template <typename F>
static int hsk_dispatch_dim(int64_t D, F&& f) {
  int V = (D % 4 == 0) ? 4 : (D % 2 == 0) ? 2 : 1;
  int64_t chunks = hsk_ceil_div(D, (int64_t)64 * V);
  int nch = chunks <= 1 ? 1 : chunks <= 2 ? 2 : chunks <= 4 ? 4 : chunks <= 8 ? 8 : 0;
  if (nch == 0) {
    hsk_set_error("embedding_dim %lld not supported (max %d for this alignment)", (long long)D, 64 * V * 8);
    return HSK_ERR_UNSUPPORTED;
  }
  bool full = (D == (int64_t)64 * V * nch);
#define HSK_CASE(v, n)                                                                          \
  if (V == v && nch == n) {                                                                     \
    if (v == 4 && full)                                                                         \
      return f(std::integral_constant<int, v>{}, std::integral_constant<int, n>{}, std::true_type{}); \
    return f(std::integral_constant<int, v>{}, std::integral_constant<int, n>{}, std::false_type{});  \
  }
  HSK_CASE(4, 1) HSK_CASE(4, 2) HSK_CASE(4, 4) HSK_CASE(4, 8)
  HSK_CASE(2, 1) HSK_CASE(2, 2) HSK_CASE(2, 4) HSK_CASE(2, 8)
  HSK_CASE(1, 1) HSK_CASE(1, 2) HSK_CASE(1, 4) HSK_CASE(1, 8)
#undef HSK_CASE
  hsk_set_error("internal: no kernel for dim %lld", (long long)D);
  return HSK_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// row helpers
// ---------------------------------------------------------------------------------------------
template <int V, int NCH>
struct hsk_row {
  hsk_vec<V> c[NCH];
};

template <int V, int NCH, bool FULL>
__device__ __forceinline__ void hsk_row_load(hsk_row<V, NCH>& r, const float* __restrict__ base, int lane, int D) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int off = (c * 64 + lane) * V;
    if (FULL || off < D)
      r.c[c] = hsk_ldg<V>(base + off);
    else
      r.c[c] = hsk_zero<V>();
  }
}

template <int V, int NCH, bool FULL>
__device__ __forceinline__ void hsk_row_store(const hsk_row<V, NCH>& r, float* __restrict__ base, int lane, int D) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int off = (c * 64 + lane) * V;
    if (FULL || off < D) hsk_stg<V>(base + off, r.c[c]);
  }
}

template <int V, int NCH>
__device__ __forceinline__ void hsk_row_zero(hsk_row<V, NCH>& r) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) r.c[c] = hsk_zero<V>();
}

template <int V, int NCH>
__device__ __forceinline__ float hsk_row_dot_partial(const hsk_row<V, NCH>& a, const hsk_row<V, NCH>& b) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < V; ++i) s = fmaf(a.c[c].v[i], b.c[c].v[i], s);
  return s;
}

template <int V, int NCH>
__device__ __forceinline__ void hsk_row_axpy(hsk_row<V, NCH>& acc, float a, const hsk_row<V, NCH>& x) {
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < V; ++i) acc.c[c].v[i] = fmaf(a, x.c[c].v[i], acc.c[c].v[i]);
}

template <int V, int NCH>
__device__ __forceinline__ void hsk_row_add(hsk_row<V, NCH>& acc, const hsk_row<V, NCH>& x) {
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < V; ++i) acc.c[c].v[i] += x.c[c].v[i];
}

__device__ __forceinline__ float hsk_softplus(float z) {
  // BCEWithLogits(x, 1) = softplus(-x) = max(-x,0) + log1p(exp(-|x|)); here z = -x
  return fmaxf(z, 0.f) + log1pf(expf(-fabsf(z)));
}

__device__ __forceinline__ int hsk_clamp_index(long long idx, long long n, int32_t* status) {
  if (idx < 0 || idx >= n) {
    if (status) atomicOr(status, HSK_STATUS_BAD_INDEX);
    return 0;
  }
  return (int)idx;
}
