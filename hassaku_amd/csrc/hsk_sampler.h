// hsk_sampler.h -- batch preparation: int32 copies of a loader batch, or on-device batch construction with the
// uniform rejection sampler (data/dataloader.py:56-57,92-129 of the reference), plus the per-user owner map.
#pragma once
#include "hsk_rows.h"

// =============================================================================================
// P0a: external batch -> int32 workspace copies, per-item histogram, per-user owner/count
// =============================================================================================
__global__ __launch_bounds__(256) void k_prep_external(const int64_t* __restrict__ u_idx,
                                                       const int64_t* __restrict__ i_idx, int B, int K,
                                                       int n_users, int n_items, int* __restrict__ u32,
                                                       int* __restrict__ it32,
                                                       int* __restrict__ owner, int* __restrict__ cnt,
                                                       int32_t* status, int* __restrict__ stamp = nullptr,
                                                       int stamp_val = 0, int n_part = 1) {
  // n_part > 1 (item-partitioned forward, hsk_fwd_part.h): rows of K + n_part - 1 columns; column 0 is the positive item,
  // columns 1 .. n_part-1 hold -1 (no entry: the item sort skips them) -- unit q of the forward leaves its share of
  // d loss/d s_0 in column q of g_s, the item pass adds the n_part shares when it meets the positive's entry
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * K;
  if (e < total) {
    const int it = hsk_clamp_index(i_idx[e], n_items, status);
    if (n_part > 1) {
      const long long b = e / K, k = e - b * K;
      int* row = it32 + b * (K + n_part - 1);
      if (k == 0)
        for (int q = 0; q < n_part; ++q) row[q] = q ? -1 : it;
      else
        row[k + n_part - 1] = it;
    } else {
      it32[e] = it;
    }
  }
  if (e < B) {
    const int u = hsk_clamp_index(u_idx[e], n_users, status);
    u32[e] = u;
    atomicMin(&owner[u], (int)e);
    atomicAdd(&cnt[u], 1);
    if (stamp) stamp[u] = stamp_val;
  }
}

// =============================================================================================
// P0b: device batch construction + uniform rejection sampler (data/dataloader.py:92-129)
// =============================================================================================
__device__ __forceinline__ bool hsk_row_has(const int32_t* __restrict__ idx, long long lo, long long hi, int key) {
  long long l = lo, h = hi;
  while (l < h) {
    const long long mid = (l + h) >> 1;
    const int v = idx[mid];
    if (v < key)
      l = mid + 1;
    else
      h = mid;
  }
  return (l < hi) && (idx[l] == key);
}

#define HSK_SAMPLER_LDS_ROW 1024   // positives of one user staged in LDS per wave (longer rows are searched in HBM)

// One draw for slot (b, n): Philox counter = (b, n, stream_lo, (stream_hi<<16) | block), 4 attempts per
// block; exact uniform integer via Lemire's multiply-shift with rejection of the biased zone.
// With an alias table (alias_prob != NULL; Walker/Vose, built on the host from pop_distribution^squash --
// NegativeSampler._neg_sample_popular, data/dataloader.py:59-64) a draw uses two words: a uniform column and a
// uniform float deciding between the column and its alias, i.e. 2 attempts per block, same rejection rule.
struct hsk_alias {
  const float* prob;     // NULL: uniform sampling
  const int32_t* alias;
};

// bitmap (optional): the user's positives as one bit per item (LDS) -- the same test as the binary search, one read
__device__ __forceinline__ int hsk_draw_negative(const int32_t* __restrict__ csr_indices, long long row_lo,
                                                 long long row_hi, uint32_t n_items, uint32_t b, uint32_t n,
                                                 uint64_t seed, uint64_t stream_id, int32_t* status,
                                                 hsk_alias at = hsk_alias{nullptr, nullptr},
                                                 const uint32_t* __restrict__ bitmap = nullptr) {
  auto taken = [&](int cand) {
    return bitmap ? ((bitmap[cand >> 5] >> (cand & 31)) & 1u) != 0u : hsk_row_has(csr_indices, row_lo, row_hi, cand);
  };
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const uint32_t thresh = (uint32_t)(-(int32_t)n_items) % n_items;  // 2^32 mod n_items
  int last = 0;
  for (uint32_t blk = 0; blk < 4096u; ++blk) {
    hsk_u32x4 ctr;
    ctr.x = b;
    ctr.y = n;
    ctr.z = (uint32_t)stream_id;
    ctr.w = ((uint32_t)(stream_id >> 32) << 16) | (blk & 0xffffu);
    const hsk_u32x4 r = hsk_philox4x32_10(ctr, k0, k1);
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
    if (at.prob) {
#pragma unroll
      for (int a = 0; a < 4; a += 2) {
        const uint64_t m = (uint64_t)rr[a] * (uint64_t)n_items;
        if ((uint32_t)m < thresh) continue;
        const int col = (int)(m >> 32);
        const float u01 = (float)(rr[a + 1] >> 8) * (1.0f / 16777216.0f);
        const int cand = (u01 < at.prob[col]) ? col : at.alias[col];
        last = cand;
        if (!taken(cand)) return cand;
      }
      continue;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const uint64_t m = (uint64_t)rr[a] * (uint64_t)n_items;
      if ((uint32_t)m < thresh) continue;  // biased zone: costs one attempt
      const int cand = (int)(m >> 32);
      last = cand;
      if (!taken(cand)) return cand;
    }
  }
  if (status) atomicOr(status, HSK_STATUS_SAMPLER_GAVE_UP);
  return last;
}

// the draws of positive b against its user's sorted positives `set` [0, len) and the batch rows they fill
__device__ __forceinline__ void hsk_sample_draw_write(int b, int lane, const int32_t* __restrict__ set, int len, int u, int ipos,
                                                      int n_neg, int n_items, uint64_t seed, uint64_t stream_id,
                                                      int* __restrict__ u32, int* __restrict__ row, int* __restrict__ owner,
                                                      int* __restrict__ cnt, int32_t* status, int b_offset, hsk_alias at,
                                                      int* __restrict__ stamp, int n_part,
                                                      const uint32_t* __restrict__ bitmap = nullptr) {
  for (int n = lane; n < n_neg; n += 64) {
    const int neg = hsk_draw_negative(set, 0, len, (uint32_t)n_items, (uint32_t)(b + b_offset), (uint32_t)n, seed,
                                      stream_id, status, at, bitmap);
    row[n_part + n] = neg;
  }
  if (lane < n_part) row[lane] = lane ? -1 : ipos;
  if (lane == 0) {
    u32[b] = u;
    if (owner) {
      atomicMin(&owner[u], b);
      atomicAdd(&cnt[u], 1);
    }
    if (stamp) stamp[u] = (int)stream_id + 1;
  }
}

// one wave per positive (b = its batch position); lanes stride over the n_neg slots
__device__ __forceinline__ void hsk_prep_sample_body(int b, int lane, int wave, const int32_t* __restrict__ coo_user,
                                                     const int32_t* __restrict__ coo_item,
                                                     const int64_t* __restrict__ order, long long start, int n_neg,
                                                     const int64_t* __restrict__ csr_indptr,
                                                     const int32_t* __restrict__ csr_indices, int n_items, uint64_t seed,
                                                     uint64_t stream_id, int* __restrict__ u32, int* __restrict__ it32,
                                                     int* __restrict__ owner, int* __restrict__ cnt, int32_t* status,
                                                     int b_offset, hsk_alias at, int* __restrict__ stamp, int n_part) {
  const long long pos = order ? (long long)order[start + b] : (start + b);
  const int u = coo_user[pos];
  const int ipos = coo_item[pos];
  const long long lo = csr_indptr[u], hi = csr_indptr[u + 1];
  const int K = n_neg + 1;
  int* row = it32 + (long long)b * (K + n_part - 1);
  // stage the user's sorted positives in LDS: the rejection test is a ~7-step binary search per draw, and as
  // dependent global loads that chain (not the RNG) is what the sampler spends its time on
  __shared__ int32_t lds_row[4][HSK_SAMPLER_LDS_ROW];
  const int len = (int)(hi - lo);
  const bool staged = len <= HSK_SAMPLER_LDS_ROW;
  if (staged) {
    for (int j = lane; j < len; j += 64) lds_row[wave][j] = csr_indices[lo + j];
    __builtin_amdgcn_wave_barrier();
  }
  const int32_t* set = staged ? lds_row[wave] : csr_indices + lo;
  hsk_sample_draw_write(b, lane, set, len, u, ipos, n_neg, n_items, seed, stream_id, u32, row, owner, cnt, status, b_offset, at,
                        stamp, n_part);
}

__global__ __launch_bounds__(256) void k_prep_sample(const int32_t* __restrict__ coo_user,
                                                     const int32_t* __restrict__ coo_item,
                                                     const int64_t* __restrict__ order, long long start, int B,
                                                     int n_neg, const int64_t* __restrict__ csr_indptr,
                                                     const int32_t* __restrict__ csr_indices, int n_items,
                                                     uint64_t seed, uint64_t stream_id, int* __restrict__ u32,
                                                     int* __restrict__ it32,
                                                     int* __restrict__ owner, int* __restrict__ cnt,
                                                     int32_t* status, int b_offset = 0,
                                                     hsk_alias at = hsk_alias{nullptr, nullptr},
                                                     const hsk_step_desc* __restrict__ desc = nullptr, int rel = 0,
                                                     int* __restrict__ stamp = nullptr, int n_part = 1) {
  // n_part > 1: rows of K + n_part - 1 columns, the positive in column 0, -1 in columns 1 .. n_part-1 (see k_prep_external)
  // stamp (optional): stamp[u] = the (1-based) step this batch is trained on, for every user of the batch -- how a
  // kernel of that step tells the rows being updated from the rows it may bring up to date ahead of time
  // b_offset: offset added to the batch position in the RNG counter (a slice of a larger global batch).
  // owner == NULL: no owner map.  desc: graph replay, see hsk_step_desc.
  if (desc) {
    start = desc->start0 + (long long)rel * B;
    order = desc->order;
    stream_id = (uint64_t)(desc->step0 + rel);
  }
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  hsk_prep_sample_body(b, lane, wave, coo_user, coo_item, order, start, n_neg, csr_indptr, csr_indices, n_items, seed,
                       stream_id, u32, it32, owner, cnt, status, b_offset, at, stamp, n_part);
}

// The sampler as extra workgroups of another launch (the in-launch preparation pipeline of hsk_fused.hip: the batch two
// steps ahead is sampled by workgroups riding in the forward's launch).  Same body, same draws as k_prep_sample.
struct hsk_ride_sample {
  int n_blocks;   // workgroups of this phase in the launch (0: none)
  const int32_t* coo_user;
  const int32_t* coo_item;
  const int64_t* order;
  long long start;
  int B, n_neg;
  const int64_t* csr_indptr;
  const int32_t* csr_indices;
  int n_items;
  uint64_t seed, stream_id;
  int *u32, *it32, *owner, *cnt;
  int32_t* status;
  hsk_alias at;
  int* stamp;
  int n_part;
  int per_wave;   // positives per wave (>= 1)
};

// per_wave <= 64 positives per wave.  One positive per wave (k_prep_sample's form) puts 4096 waves in front of the host
// launch, each a chain of four dependent loads (order -> coo -> indptr -> the user's row) before its first draw: they hold
// EVERY wave slot of the chip for the forward's first ~6 us (measured: 6.9 us of it).  Here lane k of a wave runs that
// chain for the wave's k-th positive -- once per wave instead of once per positive -- and the next positive's row
// (its first 256 entries, four registers) is fetched while the current one draws: the same draws from a fraction of the
// wave-time, in workgroups that leave the host's own most of the slots from its first microsecond.
__device__ __forceinline__ void hsk_ride_sample_body(const hsk_ride_sample& a, int bid) {
  __shared__ int32_t ride_row[4][HSK_SAMPLER_LDS_ROW];
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int K = a.n_neg + 1;
  // lane k: the wave's k-th positive
  const int my_b = (lane * a.n_blocks + bid) * 4 + wave;
  long long my_lo = 0;
  int my_u = 0, my_ipos = 0, my_len = -1;   // (-1: no such positive)
  if (lane < a.per_wave && my_b < a.B) {
    const long long pos = a.order ? (long long)a.order[a.start + my_b] : (a.start + my_b);
    my_u = a.coo_user[pos];
    my_ipos = a.coo_item[pos];
    my_lo = a.csr_indptr[my_u];
    my_len = (int)(a.csr_indptr[my_u + 1] - my_lo);
  }
  int32_t nxt[4];
  auto fetch = [&](int k) {   // the first 256 entries of positive k's row -> nxt
    const int len = __shfl(my_len, k, 64);
    const long long lo = ((long long)__shfl((int)(my_lo >> 32), k, 64) << 32) | (uint32_t)__shfl((int)my_lo, k, 64);
#pragma unroll
    for (int j = 0; j < 4; ++j) nxt[j] = (lane + 64 * j < len) ? a.csr_indices[lo + lane + 64 * j] : 0;
  };
  // catalogues of up to 32 768 items: the wave's LDS row holds the user's positives as a BITMAP (one bit per item) -- the
  // rejection test is then one LDS read instead of a ~7-step binary search, each step an LDS round trip.  The words a
  // positive's row sets are cleared again behind its draws, so the bitmap is zeroed once per wave.
  const bool bm = a.n_items <= 32 * HSK_SAMPLER_LDS_ROW;
  uint32_t* bits = reinterpret_cast<uint32_t*>(ride_row[wave]);
  if (bm) {
    for (int j = lane; j < (a.n_items + 31) / 32; j += 64) bits[j] = 0u;
    __builtin_amdgcn_wave_barrier();
  }
  fetch(0);
  for (int k = 0; k < a.per_wave; ++k) {
    const int len = __shfl(my_len, k, 64);
    if (len < 0) return;   // wave-uniform: positives are handed out in ascending k
    const int b = (k * a.n_blocks + bid) * 4 + wave;
    const int u = __shfl(my_u, k, 64), ipos = __shfl(my_ipos, k, 64);
    const long long lo = ((long long)__shfl((int)(my_lo >> 32), k, 64) << 32) | (uint32_t)__shfl((int)my_lo, k, 64);
    int32_t cur[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
    const bool staged = !bm && len <= HSK_SAMPLER_LDS_ROW;
    if (bm) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (lane + 64 * j < len) atomicOr(&bits[cur[j] >> 5], 1u << (cur[j] & 31));
      for (int j = 256 + lane; j < len; j += 64) {
        const int it = a.csr_indices[lo + j];
        atomicOr(&bits[it >> 5], 1u << (it & 31));
      }
      __builtin_amdgcn_wave_barrier();
    } else if (staged) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (lane + 64 * j < len) ride_row[wave][lane + 64 * j] = cur[j];
      for (int j = 256 + lane; j < len; j += 64) ride_row[wave][j] = a.csr_indices[lo + j];
      __builtin_amdgcn_wave_barrier();
    }
    if (k + 1 < a.per_wave) fetch(k + 1);   // in flight under this positive's draws
    const int32_t* set = staged ? ride_row[wave] : a.csr_indices + lo;
    hsk_sample_draw_write(b, lane, set, len, u, ipos, a.n_neg, a.n_items, a.seed, a.stream_id, a.u32,
                          a.it32 + (long long)b * (K + a.n_part - 1), a.owner, a.cnt, a.status, 0, a.at, a.stamp, a.n_part,
                          bm ? bits : nullptr);
    __builtin_amdgcn_wave_barrier();   // (the wave's LDS row is reused by the next positive)
    if (bm) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (lane + 64 * j < len) bits[cur[j] >> 5] = 0u;
      for (int j = 256 + lane; j < len; j += 64) bits[a.csr_indices[lo + j] >> 5] = 0u;
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// The batches of n_group consecutive steps of a replayed run in ONE launch (graph replay only: step / start / order come
// from the descriptor): batch g = relative step rel0 + g, written to slot g of the per-batch buffers (gs_* = distance
// between the slots, in elements).  Same draws as n_group launches of k_prep_sample.
__global__ __launch_bounds__(256) void k_prep_sample_group(const int32_t* __restrict__ coo_user,
                                                           const int32_t* __restrict__ coo_item, int B, int n_neg,
                                                           const int64_t* __restrict__ csr_indptr,
                                                           const int32_t* __restrict__ csr_indices, int n_items,
                                                           uint64_t seed, int* __restrict__ u32, int* __restrict__ it32,
                                                           int* __restrict__ owner, int* __restrict__ cnt, int32_t* status,
                                                           hsk_alias at, const hsk_step_desc* __restrict__ desc, int rel0,
                                                           int* __restrict__ stamp, int n_group, long long gs_batch,
                                                           long long gs_ent, long long gs_users) {
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int bb = blockIdx.x * 4 + wave;
  const int g = bb / B, b = bb - g * B;
  if (g >= n_group) return;
  const int rel = rel0 + g;
  hsk_prep_sample_body(b, lane, wave, coo_user, coo_item, desc->order, desc->start0 + (long long)rel * B, n_neg, csr_indptr,
                       csr_indices, n_items, seed, (uint64_t)(desc->step0 + rel), u32 + g * gs_batch, it32 + g * gs_ent,
                       owner + g * gs_users, cnt + g * gs_users, status, 0, at, stamp + g * gs_users, 1);
}

// stand-alone sampler on the int64 drop-in surface
__global__ __launch_bounds__(256) void k_sample_negatives(const int64_t* __restrict__ csr_indptr,
                                                          const int32_t* __restrict__ csr_indices, int n_users,
                                                          int n_items, const int64_t* __restrict__ u_idx, int B,
                                                          int n_neg, uint64_t seed, uint64_t stream_id,
                                                          int64_t* __restrict__ out, int32_t* status,
                                                          hsk_alias at = hsk_alias{nullptr, nullptr}) {
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const int u = hsk_clamp_index(u_idx[b], n_users, status);
  const long long lo = csr_indptr[u], hi = csr_indptr[u + 1];
  for (int n = lane; n < n_neg; n += 64)
    out[(long long)b * n_neg + n] = hsk_draw_negative(csr_indices, lo, hi, (uint32_t)n_items, (uint32_t)b,
                                                      (uint32_t)n, seed, stream_id, status, at);
}
