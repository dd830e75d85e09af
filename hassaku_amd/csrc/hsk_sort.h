// hsk_sort.h -- deterministic, atomic-free grouping of the batch entries e = b*K + k by item id.
//
// The item-major gradient pass needs, per item, the list of entries that reference it.  A two-level
// stable counting sort builds it (no global atomics, bitwise reproducible: inside an item the entries stay
// in ascending e, so the fp32 summation order of an item-row gradient is a function of the batch only):
//
//   level 1  bucket = item >> shift (NB <= 512 buckets).  The entry range is cut into NU <= 512 wave-units
//            of `epw` consecutive entries.
//     k_sort_hist     per unit: histogram of its buckets in LDS           -> hist[bucket][unit]
//     k_sort_rowscan  per bucket (one wave): exclusive scan over the units -> hist in place, btot[bucket]
//     k_sort_scatter  bucket starts = scan of btot (every workgroup, in LDS); per unit, 64 entries at a time
//                     in order: rank inside the chunk by a ballot "match-any", position = bucket start +
//                     running base of (bucket, unit) + rank                 -> perm1, bstart
//   level 2  k_sort_bucket: one workgroup per bucket, each wave owns a contiguous quarter of the bucket's
//            entries: count per (wave, item), prefix, then the same ordered placement -> perm, offsets.
// Every wave issues its index loads for 16 chunks (1024 entries) before it consumes them: these kernels are
// latency-bound, not bandwidth-bound (the whole entry list is 1.6 MB at the ml10m shape).
#pragma once
#include "hsk_common.h"
#if defined(__HIPCC__)
#include <rocprim/block/block_radix_sort.hpp>
#endif

#define HSK_SORT_MAX_BUCKETS 512
#define HSK_SORT_MAX_UNITS 512
#define HSK_SORT_MAX_IPB 8192  // items per bucket the level-2 LDS counters can hold (5 x IPB ints <= 160 KB)
#define HSK_SORT_GROUP 16      // chunks of 64 entries whose loads are issued together

struct hsk_sort_plan {
  int shift;      // bucket = item >> shift
  int n_buckets;  // NB
  int ipb;        // 1 << shift
  int epw;        // entries per wave-unit (multiple of 1024)
  int n_units;    // NU
};

static inline int hsk_make_sort_plan(int64_t n_items, int64_t n_entries, hsk_sort_plan* p) {
  int shift = 0;
  while (((n_items - 1) >> shift) + 1 > HSK_SORT_MAX_BUCKETS) ++shift;
  p->shift = shift;
  p->ipb = 1 << shift;
  p->n_buckets = (int)(((n_items - 1) >> shift) + 1);
  if (p->ipb > HSK_SORT_MAX_IPB) return -1;
  int64_t epw = 1024;
  while (hsk_ceil_div(n_entries, epw) > HSK_SORT_MAX_UNITS) epw *= 2;
  p->epw = (int)epw;
  p->n_units = (int)hsk_ceil_div(n_entries, epw);
  return 0;
}

static inline int64_t hsk_sort_hist_elems(int64_t n_items, int64_t max_entries) {
  hsk_sort_plan p;
  if (hsk_make_sort_plan(n_items, max_entries, &p) != 0) return -1;
  return (int64_t)p.n_buckets * HSK_SORT_MAX_UNITS;
}

#if defined(__HIPCC__)

// lanes holding the same key as this lane (among `valid` lanes); keys < 2^nbits
__device__ __forceinline__ unsigned long long hsk_match_any(int key, int nbits, bool valid) {
  unsigned long long mask = __ballot(valid);
  for (int bit = 0; bit < nbits; ++bit) {
    const bool one = (key >> bit) & 1;
    const unsigned long long bm = __ballot(one);
    mask &= one ? bm : ~bm;
  }
  return mask;
}

__device__ __forceinline__ int hsk_bits_for(int n) {  // smallest nbits with 2^nbits >= n
  int b = 0;
  while ((1 << b) < n) ++b;
  return b;
}

__device__ __forceinline__ int hsk_wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

// n_dev (optional, every sort kernel): the actual entry count lives in device memory (item-sharded step: how many
// of the drawn negatives a rank owns is only known on the device); n_entries is then the host-side capacity
__device__ __forceinline__ int hsk_sort_count(int n_entries, const int* __restrict__ n_dev) {
  return n_dev ? min(n_entries, *n_dev) : n_entries;
}

// Bodies + thin kernels: the same bodies also run as extra workgroups of the step's two big launches (the in-launch
// preparation pipeline of hsk_fused.hip: `bid` = the workgroup's index inside its phase, LDS handed in by the caller).
#define HSK_SORT_HIST_LDS (4 * HSK_SORT_MAX_BUCKETS)                             // ints of LDS of the histogram phase
#define HSK_SORT_SCATTER_LDS (4 * HSK_SORT_MAX_BUCKETS + HSK_SORT_MAX_BUCKETS + 1)   // ... of the scatter phase
__device__ __forceinline__ void hsk_sort_hist_body(const int* __restrict__ it32, int n_entries, const hsk_sort_plan& p,
                                                   int* __restrict__ hist, int bid, int* __restrict__ cnt_lds) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int* cnt = cnt_lds + w * HSK_SORT_MAX_BUCKETS;
  const int unit = bid * 4 + w;
  const int lo = unit * p.epw, hi = unit < p.n_units ? min(n_entries, lo + p.epw) : lo;
  // (the unit's first 1024 entries are requested before the counters are cleared: one memory round trip, not two)
  int key[HSK_SORT_GROUP];
#pragma unroll
  for (int j = 0; j < HSK_SORT_GROUP; ++j) {
    const int e = lo + j * 64 + lane;
    key[j] = (e < hi) ? (it32[e] >> p.shift) : -1;
  }
  for (int d = lane; d < p.n_buckets; d += 64) cnt[d] = 0;
  __syncthreads();
  for (int g0 = lo; g0 < hi; g0 += 64 * HSK_SORT_GROUP) {
    if (g0 != lo) {
#pragma unroll
      for (int j = 0; j < HSK_SORT_GROUP; ++j) {
        const int e = g0 + j * 64 + lane;
        key[j] = (e < hi) ? (it32[e] >> p.shift) : -1;
      }
    }
#pragma unroll
    for (int j = 0; j < HSK_SORT_GROUP; ++j)
      if (key[j] >= 0) atomicAdd(&cnt[key[j]], 1);
  }
  __syncthreads();
  if (unit < p.n_units)
    for (int d = lane; d < p.n_buckets; d += 64) hist[d * p.n_units + unit] = cnt[d];
}

__global__ __launch_bounds__(256) void k_sort_hist(const int* __restrict__ it32, int n_entries, hsk_sort_plan p,
                                                   int* __restrict__ hist, const int* __restrict__ n_dev = nullptr) {
  __shared__ int cnt[HSK_SORT_HIST_LDS];
  hsk_sort_hist_body(it32, hsk_sort_count(n_entries, n_dev), p, hist, (int)blockIdx.x, cnt);
}

// one wave per bucket: exclusive scan of hist[d][0..NU) in place, btot[d] = row total
__device__ __forceinline__ void hsk_sort_rowscan_body(int* __restrict__ hist, const hsk_sort_plan& p,
                                                      int* __restrict__ btot, int bid) {
  const int lane = threadIdx.x & 63;
  const int d = bid * 4 + (threadIdx.x >> 6);
  if (d >= p.n_buckets) return;
  int* row = hist + (long long)d * p.n_units;
  int v[HSK_SORT_MAX_UNITS / 64];
#pragma unroll
  for (int j = 0; j < HSK_SORT_MAX_UNITS / 64; ++j) {
    const int u = j * 64 + lane;
    v[j] = (u < p.n_units) ? row[u] : 0;
  }
  int carry = 0;
#pragma unroll
  for (int j = 0; j < HSK_SORT_MAX_UNITS / 64; ++j) {
    const int u = j * 64 + lane;
    const int incl = hsk_wave_incl_scan(v[j], lane);
    if (u < p.n_units) row[u] = carry + incl - v[j];
    carry += __shfl(incl, 63, 64);
  }
  if (lane == 0) btot[d] = carry;
}

__global__ __launch_bounds__(256) void k_sort_rowscan(int* __restrict__ hist, hsk_sort_plan p, int* __restrict__ btot) {
  hsk_sort_rowscan_body(hist, p, btot, (int)blockIdx.x);
}

// exclusive scan of btot[0..NB) into LDS bs[0..NB] by the calling workgroup's wave 0 (NB <= 512)
__device__ __forceinline__ void hsk_bucket_starts(const int* __restrict__ btot, int nb, int* bs) {
  const int lane = threadIdx.x & 63;
  if ((threadIdx.x >> 6) == 0) {
    // every total loaded before the first is scanned (a load per loop iteration made the head of every scatter
    // workgroup a chain of up to 8 memory latencies)
    int v[HSK_SORT_MAX_BUCKETS / 64];
#pragma unroll
    for (int q = 0; q < HSK_SORT_MAX_BUCKETS / 64; ++q) {
      const int j = q * 64 + lane;
      v[q] = (j < nb) ? btot[j] : 0;
    }
    int carry = 0;
#pragma unroll
    for (int q = 0; q < HSK_SORT_MAX_BUCKETS / 64; ++q) {
      const int j = q * 64 + lane;
      if (q * 64 < nb) {   // wave-uniform
        const int incl = hsk_wave_incl_scan(v[q], lane);
        if (j < nb) bs[j] = carry + incl - v[q];
        carry += __shfl(incl, 63, 64);
      }
    }
    if (lane == 0) bs[nb] = carry;
  }
  __syncthreads();
}

__device__ __forceinline__ void hsk_sort_scatter_body(const int* __restrict__ it32, int n_entries, const hsk_sort_plan& p,
                                                      const int* __restrict__ hist, const int* __restrict__ btot,
                                                      int2* __restrict__ perm1, int* __restrict__ bstart, int bid,
                                                      int* __restrict__ lds /* HSK_SORT_SCATTER_LDS ints */) {
  int* bs = lds + 4 * HSK_SORT_MAX_BUCKETS;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int* run = lds + w * HSK_SORT_MAX_BUCKETS;
  const int unit = bid * 4 + w;
  const bool live = unit < p.n_units;
  const int lo = unit * p.epw, hi = live ? min(n_entries, lo + p.epw) : lo;
  // every global load of the unit is issued BEFORE the scan of the bucket totals and its barrier: the unit's column of
  // the histogram and its first 1024 entries (all of them at epw = 1024).  Behind the barrier they were two more memory
  // round trips in a row, and a workgroup that rides in another launch holds its slots -- and slows its host -- for as
  // long as it lives
  int hv[HSK_SORT_MAX_BUCKETS / 64];
#pragma unroll
  for (int q = 0; q < HSK_SORT_MAX_BUCKETS / 64; ++q) {
    const int d = q * 64 + lane;
    hv[q] = (live && d < p.n_buckets) ? hist[d * p.n_units + unit] : 0;
  }
  int itemv[HSK_SORT_GROUP];
#pragma unroll
  for (int j = 0; j < HSK_SORT_GROUP; ++j) {
    const int e = lo + j * 64 + lane;
    itemv[j] = (e < hi) ? it32[e] : -1;
  }
  hsk_bucket_starts(btot, p.n_buckets, bs);
  if (bid == 0)
    for (int d = threadIdx.x; d <= p.n_buckets; d += 256) bstart[d] = bs[d];
  if (!live) return;  // no barrier below: every wave works on its own LDS row
#pragma unroll
  for (int q = 0; q < HSK_SORT_MAX_BUCKETS / 64; ++q) {
    const int d = q * 64 + lane;
    if (d < p.n_buckets) run[d] = bs[d] + hv[q];
  }
  __builtin_amdgcn_wave_barrier();
  const int nbits = hsk_bits_for(p.n_buckets);
  for (int g0 = lo; g0 < hi; g0 += 64 * HSK_SORT_GROUP) {
    if (g0 != lo) {
#pragma unroll
      for (int j = 0; j < HSK_SORT_GROUP; ++j) {
        const int e = g0 + j * 64 + lane;
        itemv[j] = (e < hi) ? it32[e] : -1;
      }
    }
#pragma unroll
    for (int j = 0; j < HSK_SORT_GROUP; ++j) {
      if (g0 + j * 64 >= hi) break;  // wave-uniform
      const bool valid = itemv[j] >= 0;
      const int key = valid ? (itemv[j] >> p.shift) : 0;
      const unsigned long long same = hsk_match_any(key, nbits, valid);
      const int rank = __popcll(same & ((1ull << lane) - 1ull));
      const int base = run[key];
      if (valid) perm1[base + rank] = make_int2(g0 + j * 64 + lane, itemv[j]);  // (entry, item)
      __builtin_amdgcn_wave_barrier();
      if (valid && rank == 0) run[key] = base + __popcll(same);  // LDS ops of one wave execute in order
      __builtin_amdgcn_wave_barrier();
    }
  }
}

__global__ __launch_bounds__(256) void k_sort_scatter(const int* __restrict__ it32, int n_entries, hsk_sort_plan p,
                                                      const int* __restrict__ hist, const int* __restrict__ btot,
                                                      int2* __restrict__ perm1, int* __restrict__ bstart,
                                                      const int* __restrict__ n_dev = nullptr) {
  __shared__ int lds[HSK_SORT_SCATTER_LDS];
  hsk_sort_scatter_body(it32, hsk_sort_count(n_entries, n_dev), p, hist, btot, perm1, bstart, (int)blockIdx.x, lds);
}

// lds: (5 * ipb + 2) ints, + ipb with `touched`
__device__ __forceinline__ void hsk_sort_bucket_body(const int2* __restrict__ perm1, int n_entries, int n_items,
                                                     const hsk_sort_plan& p, const int* __restrict__ bstart,
                                                     int* __restrict__ perm, int* __restrict__ offsets,
                                                     int* __restrict__ touched, int* __restrict__ n_touched, int bid,
                                                     int* __restrict__ lds) {
  // touched / n_touched (optional): compact list of the items that have entries, any order (lazy item AdamW:
  // hsk_fused.hip); needs ipb + 2 more ints of LDS
  // lds: cnt[4][ipb] then tot[ipb] [then list[ipb], count, base]
  const int ipb = p.ipb;
  int* cnt = lds;
  int* tot = lds + 4 * ipb;
  int* tlist = lds + 5 * ipb;
  int* tmeta = lds + 6 * ipb;   // [0] items with entries in this bucket, [1] their base in `touched`
  if (touched && threadIdx.x == 0) tmeta[0] = 0;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = bid;
  const int beg = bstart[b], end = bstart[b + 1];
  const int n = end - beg;
  const int per = ((n + 3) / 4 + 63) / 64 * 64;  // entries per wave, multiple of 64 so chunks stay aligned
  const int wlo = beg + min(n, w * per), whi = beg + min(n, (w + 1) * per);
  const int item0 = b << p.shift;
  // the first HSK_SORT_GROUP chunks of this wave stay in registers for both passes (the usual case: all of them)
  int2 first[HSK_SORT_GROUP];
#pragma unroll
  for (int j = 0; j < HSK_SORT_GROUP; ++j) {
    const int q = wlo + j * 64 + lane;
    first[j] = (q < whi) ? perm1[q] : make_int2(-1, -1);
  }
  for (int j = tid; j < 4 * ipb; j += 256) cnt[j] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < HSK_SORT_GROUP; ++j)
    if (first[j].x >= 0) atomicAdd(&cnt[w * ipb + (first[j].y - item0)], 1);
  for (int q = wlo + 64 * HSK_SORT_GROUP + lane; q < whi; q += 64) atomicAdd(&cnt[w * ipb + (perm1[q].y - item0)], 1);
  __syncthreads();
  // per item: total and per-wave exclusive bases (local, relative to the bucket start)
  for (int j = tid; j < ipb; j += 256) {
    const int c0 = cnt[j], c1 = cnt[ipb + j], c2 = cnt[2 * ipb + j], c3 = cnt[3 * ipb + j];
    tot[j] = c0 + c1 + c2 + c3;
    if (touched && c0 + c1 + c2 + c3 > 0) tlist[atomicAdd(&tmeta[0], 1)] = item0 + j;
    cnt[j] = 0;
    cnt[ipb + j] = c0;
    cnt[2 * ipb + j] = c0 + c1;
    cnt[3 * ipb + j] = c0 + c1 + c2;
  }
  __syncthreads();
  // exclusive scan of tot over the bucket's items (wave 0, sequential over chunks of 64)
  if (w == 0) {
    int carry = 0;
    for (int j0 = 0; j0 < ipb; j0 += 64) {
      const int j = j0 + lane;
      const int v = (j < ipb) ? tot[j] : 0;
      const int incl = hsk_wave_incl_scan(v, lane);
      if (j < ipb) tot[j] = carry + incl - v;
      carry += __shfl(incl, 63, 64);
    }
  }
  __syncthreads();
  for (int j = tid; j < ipb; j += 256) {
    const int start = beg + tot[j];
    if (item0 + j < n_items) offsets[item0 + j] = start;
    cnt[j] += start;
    cnt[ipb + j] += start;
    cnt[2 * ipb + j] += start;
    cnt[3 * ipb + j] += start;
  }
  if (b == p.n_buckets - 1 && tid == 0) offsets[n_items] = end;   // = the entries with an item id (negative ids: no entry)
  if (touched && tid == 0) tmeta[1] = atomicAdd(n_touched, tmeta[0]);
  __syncthreads();
  if (touched)
    for (int j = tid; j < tmeta[0]; j += 256) touched[tmeta[1] + j] = tlist[j];
  const int nbits = hsk_bits_for(ipb);
  int* run = cnt + w * ipb;
  auto place = [&](int2 ent) {   // one chunk of 64 entries, in order
    const bool valid = ent.x >= 0;
    const int key = valid ? (ent.y - item0) : 0;
    const unsigned long long same = hsk_match_any(key, nbits, valid);
    const int rank = __popcll(same & ((1ull << lane) - 1ull));
    const int base = run[key];
    if (valid) perm[base + rank] = ent.x;
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0) run[key] = base + __popcll(same);
    __builtin_amdgcn_wave_barrier();
  };
#pragma unroll
  for (int j = 0; j < HSK_SORT_GROUP; ++j) {
    if (wlo + j * 64 >= whi) break;  // wave-uniform
    place(first[j]);
  }
  for (int c = wlo + 64 * HSK_SORT_GROUP; c < whi; c += 64) {
    const int q = c + lane;
    place((q < whi) ? perm1[q] : make_int2(-1, -1));
  }
}

__global__ __launch_bounds__(256) void k_sort_bucket(const int2* __restrict__ perm1, int n_entries, int n_items,
                                                     hsk_sort_plan p, const int* __restrict__ bstart,
                                                     int* __restrict__ perm, int* __restrict__ offsets,
                                                     int* __restrict__ touched = nullptr,
                                                     int* __restrict__ n_touched = nullptr,
                                                     const int* __restrict__ n_dev = nullptr) {
  extern __shared__ int lds[];
  hsk_sort_bucket_body(perm1, hsk_sort_count(n_entries, n_dev), n_items, p, bstart, perm, offsets, touched, n_touched,
                       (int)blockIdx.x, lds);
}


// One phase of the two-level sort as extra workgroups of another launch (the in-launch preparation pipeline of
// hsk_fused.hip): which phase is said by where the struct sits (hsk_ride_fwd / hsk_ride_item).
#define HSK_PIPE_MAX_IPB 128   // items per bucket the riding bucket phase holds counters for (5 * ipb + 2 ints of LDS)
struct hsk_ride_sort {
  int n_blocks;   // workgroups of this phase in the launch (0: none)
  const int* it32;
  int n_entries, n_items;
  hsk_sort_plan plan;
  int* hist;
  int* btot;
  int2* perm1;
  int* bstart;
  int* perm;
  int* offsets;
};

// ---------------------------------------------------------------------------------------------
// Small batches (<= 1024 * IPT entries): the whole sort in ONE workgroup.  At B = 128, N = 50 the four level-1/2
// kernels above are 43 us of launch latencies for 6 528 entries -- longer than the rest of the step.  A stable block
// radix sort (rocPRIM's block primitive; blocked arrangement, so equal items keep ascending entry order) of
// (item, entry) gives perm directly; offsets come from the run boundaries of the sorted keys.
// ---------------------------------------------------------------------------------------------
template <int IPT>
__global__ __launch_bounds__(1024) void k_sort_small(const int* __restrict__ it32, int n_entries, int n_items,
                                                     int nbits, int* __restrict__ perm, int* __restrict__ offsets,
                                                     int* __restrict__ touched = nullptr,
                                                     int* __restrict__ n_touched = nullptr,
                                                     const int* __restrict__ n_dev = nullptr) {
  n_entries = hsk_sort_count(n_entries, n_dev);
  using sort_t = rocprim::block_radix_sort<unsigned int, 1024, IPT, int>;
  __shared__ typename sort_t::storage_type storage;
  const int tid = threadIdx.x;
  unsigned int keys[IPT];
  int vals[IPT];
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int e = tid * IPT + j;
    keys[j] = (e < n_entries) ? (unsigned int)it32[e] : (unsigned int)n_items;   // pads sort behind every item
    vals[j] = e;
  }
  sort_t().sort(keys, vals, storage, 0, nbits);
  __shared__ unsigned int last_key[1024];
  __shared__ int nt;
  if (tid == 0) nt = 0;
  last_key[tid] = keys[IPT - 1];
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int pos = tid * IPT + j;
    if (pos < n_entries) perm[pos] = vals[j];
  }
  __syncthreads();
  // offsets from the run boundaries of the sorted keys (in registers): where the key steps from p to c at position
  // pos, every item in (p, c] starts at pos (items without entries included).  The first padding key (n_items, at
  // position n_entries) closes the list: offsets[n_items] = n_entries.
  int prev = tid ? (int)last_key[tid - 1] : -1;
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int pos = tid * IPT + j;
    const int cur = (int)keys[j];
    for (int i = prev + 1; i <= cur; ++i) offsets[i] = pos;
    if (touched && cur != prev && cur < n_items) touched[atomicAdd(&nt, 1)] = cur;   // first entry of item `cur`
    prev = cur;
  }
  if (touched) {
    __syncthreads();
    if (tid == 0) *n_touched = nt;
  }
  if (tid == 1023 && n_entries == 1024 * IPT)   // no padding key: close the list here
    for (int i = prev + 1; i <= n_items; ++i) offsets[i] = n_entries;
}

// ---------------------------------------------------------------------------------------------
// Small batches over catalogues with few entries per item (<= 8192 entries, on average <= 4 per item): ONE workgroup
// of 1024 threads, everything in LDS.  The block radix sort above spends ~34 us on 6 528 entries of 12-bit keys
// (BASELINE configs[1]) -- by then the longest kernel of the step -- because every one of its passes moves every entry.
// Here:   count per item (LDS atomics)  ->  scan over the items (16 waves)  ->  scatter through per-item LDS cursors.
// The cursors hand out positions in whatever order the atomics land, so each item's short list is then put into
// ascending entry order: lists of <= 8 entries by their own thread (registers, a fixed compare-exchange network),
// longer ones by a wave (rank = number of smaller entries, counted from LDS).  The result is the stable sort by item,
// bit for bit what the other two sorts produce.  LDS: 3 * n_items + 2 * 8192 + 64 ints (the second 8192 holds the
// work list of long items; sized for the worst case).
// ---------------------------------------------------------------------------------------------
#define HSK_SORT_LDS_CAP 8192
static inline size_t hsk_sort_lds_bytes(int64_t n_items) { return ((size_t)3 * n_items + 2 * HSK_SORT_LDS_CAP + 64) * sizeof(int); }
#define HSK_SORT_LDS_MAX_ITEMS 8000   // 3 x n_items + 16 448 ints <= 160 KB
// few entries per item on average: the per-item fix-up is what this sort adds to a plain scatter
static inline bool hsk_sort_lds_fits(int64_t n_items, int64_t n_entries) {
  return n_entries <= HSK_SORT_LDS_CAP && n_items <= HSK_SORT_LDS_MAX_ITEMS && n_entries <= 4 * n_items;
}

__device__ __forceinline__ void hsk_cswap(int& a, int& b) {
  const int lo = min(a, b), hi = max(a, b);
  a = lo;
  b = hi;
}

__global__ __launch_bounds__(1024) void k_sort_lds(const int* __restrict__ it32, int n_entries, int n_items,
                                                   int* __restrict__ perm, int* __restrict__ offsets,
                                                   int* __restrict__ touched = nullptr,
                                                   int* __restrict__ n_touched = nullptr,
                                                   const int* __restrict__ n_dev = nullptr, long long gs_ent = 0,
                                                   long long gs_items = 0) {
  // grouped preparation: workgroup g sorts the batch in slot g of the per-batch buffers (one workgroup: g = 0)
  it32 += blockIdx.x * gs_ent;
  perm += blockIdx.x * gs_ent;
  offsets += blockIdx.x * gs_items;
  n_entries = hsk_sort_count(n_entries, n_dev);
  extern __shared__ int lds[];
  const int I = n_items;
  int* cnt = lds;                       // [I]  entries per item
  int* beg = lds + I;                   // [I]  first position of the item's list
  int* cur = lds + 2 * I;               // [I]  scatter cursor
  int* tmp = lds + 3 * I;               // [CAP] entries, grouped by item
  int* tmp2 = tmp + HSK_SORT_LDS_CAP;   // [CAP] items whose list is longer than 8 entries
  int* meta = tmp2 + HSK_SORT_LDS_CAP;  // [0..15] wave totals, [16] touched count, [17] long-list count
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  constexpr int IPT = HSK_SORT_LDS_CAP / 1024;
  int keys[IPT];
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int e = j * 1024 + tid;
    keys[j] = (e < n_entries) ? it32[e] : -1;
  }
  for (int j = tid; j < I; j += 1024) cnt[j] = 0;
  if (tid < 64) meta[tid] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < IPT; ++j)
    if (keys[j] >= 0) atomicAdd(&cnt[keys[j]], 1);
  __syncthreads();
  // exclusive scan of cnt over the items: wave w owns a contiguous run of whole 64-item chunks
  const int chunks = (I + 63) / 64, cpw = (chunks + 15) / 16;
  const int c_lo = min(chunks, w * cpw), c_hi = min(chunks, (w + 1) * cpw);
  int carry = 0;
  for (int c = c_lo; c < c_hi; ++c) {
    const int j = c * 64 + lane;
    const int v = (j < I) ? cnt[j] : 0;
    const int incl = hsk_wave_incl_scan(v, lane);
    if (j < I) beg[j] = carry + incl - v;
    carry += __shfl(incl, 63, 64);
  }
  if (lane == 0) meta[w] = carry;
  __syncthreads();
  int wbase = 0;
  for (int ww = 0; ww < w; ++ww) wbase += meta[ww];
  for (int c = c_lo; c < c_hi; ++c) {
    const int j = c * 64 + lane;
    if (j < I) {
      const int st = beg[j] + wbase;
      beg[j] = st;
      cur[j] = st;
      offsets[j] = st;
      if (touched && cnt[j] > 0) touched[atomicAdd(&meta[16], 1)] = j;
    }
  }
  if (tid == 0) offsets[I] = n_entries;
  __syncthreads();
  if (touched && tid == 0) *n_touched = meta[16];
#pragma unroll
  for (int j = 0; j < IPT; ++j)
    if (keys[j] >= 0) tmp[atomicAdd(&cur[keys[j]], 1)] = j * 1024 + tid;
  __syncthreads();
  // short lists: one thread each, in registers
  for (int j = tid; j < I; j += 1024) {
    const int n = cnt[j];
    if (n == 0) continue;
    if (n > 8) {
      tmp2[atomicAdd(&meta[17], 1)] = j;   // work list of the long-list pass (at most CAP/9 items)
      continue;
    }
    const int st = beg[j];
    int v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (q < n) ? tmp[st + q] : 0x7fffffff;
    // optimal 19-exchange network for 8 keys
    hsk_cswap(v[0], v[2]); hsk_cswap(v[1], v[3]); hsk_cswap(v[4], v[6]); hsk_cswap(v[5], v[7]);
    hsk_cswap(v[0], v[4]); hsk_cswap(v[1], v[5]); hsk_cswap(v[2], v[6]); hsk_cswap(v[3], v[7]);
    hsk_cswap(v[0], v[1]); hsk_cswap(v[2], v[3]); hsk_cswap(v[4], v[5]); hsk_cswap(v[6], v[7]);
    hsk_cswap(v[2], v[4]); hsk_cswap(v[3], v[5]);
    hsk_cswap(v[1], v[4]); hsk_cswap(v[3], v[6]);
    hsk_cswap(v[1], v[2]); hsk_cswap(v[3], v[4]); hsk_cswap(v[5], v[6]);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q < n) perm[st + q] = v[q];
  }
  __syncthreads();
  // long lists: one wave each; an entry's place = number of smaller entries of its list (they are distinct), counted
  // from LDS (all lanes read the same word: a broadcast)
  const int n_long = meta[17];
  for (int t = w; t < n_long; t += 16) {
    const int j = tmp2[t];
    const int st = beg[j], n = cnt[j];
    for (int i0 = 0; i0 < n; i0 += 64) {
      const int i = i0 + lane;
      const int mine = (i < n) ? tmp[st + i] : 0x7fffffff;
      int rank = 0;
      for (int k = 0; k < n; ++k) rank += (tmp[st + k] < mine) ? 1 : 0;
      if (i < n) perm[st + rank] = mine;
    }
  }
}

#endif  // __HIPCC__
