// hsk_train.hip -- BPR-MF training path for gfx950 (MI355X): sampler, gather+score+loss+row-grads,
// item-major gradient reduction fused with AdamW, user-table AdamW.  Wave64 everywhere.
//
// Data layout in HBM: embedding tables are row-major fp32 [rows, D] (ld == D).  One wavefront owns
// one embedding row at a time: lane l holds elements (c*64 + l)*V .. +V of the row for chunk c, so a
// row moves as NCH coalesced global_load_dwordx{V} per lane and a dot product is NCH*V FMAs per lane
// plus one DPP wave reduction.
//
// Reference semantics restated (paths relative to the reference tree):
//   forward   algorithms/sgd_alg.py:148-179, loss train/rec_losses.py:68-88,
//   backward  autograd of the above (embedding_dense_backward), optimizer torch.optim.AdamW
//   (train/trainer.py:52-53,146-148), sampler data/dataloader.py:56-57,92-129.
#include "hsk_common.h"

#include <type_traits>

#define HSK_OWNER_NONE 0x7fffffff

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

// =============================================================================================
// P0a: external batch -> int32 workspace copies, per-item histogram, per-user owner/count
// =============================================================================================
__global__ __launch_bounds__(256) void k_prep_external(const int64_t* __restrict__ u_idx,
                                                       const int64_t* __restrict__ i_idx, int B, int K,
                                                       int n_users, int n_items, int* __restrict__ u32,
                                                       int* __restrict__ it32, int* __restrict__ counts,
                                                       int* __restrict__ owner, int* __restrict__ cnt,
                                                       int32_t* status) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * K;
  if (e < total) {
    const int it = hsk_clamp_index(i_idx[e], n_items, status);
    it32[e] = it;
    atomicAdd(&counts[it], 1);
  }
  if (e < B) {
    const int u = hsk_clamp_index(u_idx[e], n_users, status);
    u32[e] = u;
    atomicMin(&owner[u], (int)e);
    atomicAdd(&cnt[u], 1);
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

// One draw for slot (b, n): Philox counter = (b, n, stream_lo, (stream_hi<<16) | block), 4 attempts per
// block; exact uniform integer via Lemire's multiply-shift with rejection of the biased zone.
__device__ __forceinline__ int hsk_draw_negative(const int32_t* __restrict__ csr_indices, long long row_lo,
                                                 long long row_hi, uint32_t n_items, uint32_t b, uint32_t n,
                                                 uint64_t seed, uint64_t stream_id, int32_t* status) {
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
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const uint64_t m = (uint64_t)rr[a] * (uint64_t)n_items;
      if ((uint32_t)m < thresh) continue;  // biased zone: costs one attempt
      const int cand = (int)(m >> 32);
      last = cand;
      if (!hsk_row_has(csr_indices, row_lo, row_hi, cand)) return cand;
    }
  }
  if (status) atomicOr(status, HSK_STATUS_SAMPLER_GAVE_UP);
  return last;
}

// one wave per positive; lanes stride over the n_neg slots
__global__ __launch_bounds__(256) void k_prep_sample(const int32_t* __restrict__ coo_user,
                                                     const int32_t* __restrict__ coo_item,
                                                     const int64_t* __restrict__ order, long long start, int B,
                                                     int n_neg, const int64_t* __restrict__ csr_indptr,
                                                     const int32_t* __restrict__ csr_indices, int n_items,
                                                     uint64_t seed, uint64_t stream_id, int* __restrict__ u32,
                                                     int* __restrict__ it32, int* __restrict__ counts,
                                                     int* __restrict__ owner, int* __restrict__ cnt,
                                                     int32_t* status) {
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const long long pos = order ? (long long)order[start + b] : (start + b);
  const int u = coo_user[pos];
  const int ipos = coo_item[pos];
  const long long lo = csr_indptr[u], hi = csr_indptr[u + 1];
  const int K = n_neg + 1;
  int* row = it32 + (long long)b * K;
  for (int n = lane; n < n_neg; n += 64) {
    const int neg = hsk_draw_negative(csr_indices, lo, hi, (uint32_t)n_items, (uint32_t)b, (uint32_t)n, seed,
                                      stream_id, status);
    row[1 + n] = neg;
    atomicAdd(&counts[neg], 1);
  }
  if (lane == 0) {
    row[0] = ipos;
    atomicAdd(&counts[ipos], 1);
    u32[b] = u;
    atomicMin(&owner[u], b);
    atomicAdd(&cnt[u], 1);
  }
}

// stand-alone sampler on the int64 drop-in surface
__global__ __launch_bounds__(256) void k_sample_negatives(const int64_t* __restrict__ csr_indptr,
                                                          const int32_t* __restrict__ csr_indices, int n_users,
                                                          int n_items, const int64_t* __restrict__ u_idx, int B,
                                                          int n_neg, uint64_t seed, uint64_t stream_id,
                                                          int64_t* __restrict__ out, int32_t* status) {
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const int u = hsk_clamp_index(u_idx[b], n_users, status);
  const long long lo = csr_indptr[u], hi = csr_indptr[u + 1];
  for (int n = lane; n < n_neg; n += 64)
    out[(long long)b * n_neg + n] = hsk_draw_negative(csr_indices, lo, hi, (uint32_t)n_items, (uint32_t)b,
                                                      (uint32_t)n, seed, stream_id, status);
}

// =============================================================================================
// P1: exclusive scan of the per-item histogram (single workgroup; I is at most a few 100k here)
// =============================================================================================
__global__ __launch_bounds__(1024) void k_scan_counts(const int* __restrict__ counts, int n,
                                                      int* __restrict__ offsets, int* __restrict__ cursor) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = t * per, hi = min(n, lo + per);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += counts[i];
  part[t] = s;
  __syncthreads();
  // Hillis-Steele inclusive scan over 1024 partials
  for (int off = 1; off < 1024; off <<= 1) {
    int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;  // exclusive prefix of this thread's segment
  for (int i = lo; i < hi; ++i) {
    offsets[i] = run;
    cursor[i] = run;
    run += counts[i];
  }
  if (t == 1023) offsets[n] = part[1023];
}

// P2: entry e = b*K + k goes to slot cursor[item]++ of the item-major permutation
__global__ __launch_bounds__(256) void k_scatter_perm(const int* __restrict__ it32, long long total,
                                                      int* __restrict__ cursor, int* __restrict__ perm) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int pos = atomicAdd(&cursor[it32[e]], 1);
  perm[pos] = (int)e;
}

// =============================================================================================
// K1: per positive b -- gather u row + (1+N) item rows, scores, BPR loss terms, d loss/d score,
//     user-row gradient (accumulated in registers).  One wave per positive.
//     reads:  4*D*(2+N) + 4*(1+N) + 4*(2+N) bytes per positive  (the "gather+BPR step" of SURVEY 8d)
//     writes: g_s [B,1+N], dUb [B,D], loss_b [B]
// =============================================================================================
template <int V, int NCH, bool FULL, int R>
__global__ __launch_bounds__(256) void k_fwd_ugrad(const float* __restrict__ Uw, const float* __restrict__ Iw,
                                                   const float* __restrict__ Ib, const int* __restrict__ u32,
                                                   const int* __restrict__ it32, int B, int K, int D,
                                                   float inv_bn, float* __restrict__ g_s,
                                                   float* __restrict__ dUb, double* __restrict__ loss_b) {
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;

  using Row = hsk_row<V, NCH>;
  const int* __restrict__ irow = it32 + (long long)b * K;
  const int u = hsk_uniform_i(u32[b]);
  const int i0 = hsk_uniform_i(irow[0]);

  Row ur, r0, acc;
  hsk_row_load<V, NCH, FULL>(ur, Uw + (long long)u * D, lane, D);
  hsk_row_load<V, NCH, FULL>(r0, Iw + (long long)i0 * D, lane, D);
  hsk_row_zero(acc);
  const float s0 = hsk_wave_sum(hsk_row_dot_partial(ur, r0)) + (Ib ? Ib[i0] : 0.f);

  float gsum = 0.f;    // sum over negatives of sigma(-x)/(B*N)   (wave-uniform)
  double lsum = 0.0;   // per-lane partial of sum softplus(-x)

  for (int kc = 1; kc < K; kc += 64) {
    const int nr = min(64, K - kc);
    const int myidx = (lane < nr) ? irow[kc + lane] : i0;
    const float mybias = Ib ? Ib[myidx] : 0.f;
    float gv = 0.f, xv = 0.f;

    Row bufA[R], bufB[R];
    // prologue
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r < nr) hsk_row_load<V, NCH, FULL>(bufA[r], Iw + (long long)hsk_readlane_i(myidx, r) * D, lane, D);

    auto process = [&](Row(&buf)[R], int j) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (j + r < nr) {
          const float s = hsk_wave_sum(hsk_row_dot_partial(ur, buf[r])) + hsk_readlane_f(mybias, j + r);
          const float x = s0 - s;
          const float g = inv_bn / (1.f + expf(x));  // sigma(-x)/(B*N) = d loss / d s_neg
          hsk_row_axpy(acc, g, buf[r]);
          gsum += g;
          gv = (lane == j + r) ? g : gv;
          xv = (lane == j + r) ? x : xv;
        }
      }
    };
    auto prefetch = [&](Row(&buf)[R], int j) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (j + r < nr)
          hsk_row_load<V, NCH, FULL>(buf[r], Iw + (long long)hsk_readlane_i(myidx, j + r) * D, lane, D);
    };

    for (int j = 0; j < nr; j += 2 * R) {
      prefetch(bufB, j + R);
      process(bufA, j);
      prefetch(bufA, j + 2 * R);
      process(bufB, j + R);
    }
    if (lane < nr) {
      g_s[(long long)b * K + kc + lane] = gv;
      lsum += (double)hsk_softplus(-xv);
    }
  }
  // positive: d loss / d s_pos = -sum_n sigma(-x_n)/(B*N)
  const float g0 = -gsum;
  hsk_row_axpy(acc, g0, r0);
  if (lane == 0) g_s[(long long)b * K] = g0;
  hsk_row_store<V, NCH, FULL>(acc, dUb + (long long)b * D, lane, D);
  const double l = hsk_wave_sum_f64(lsum);
  if (lane == 0) loss_b[b] = l;
}

// =============================================================================================
// K2: per item i -- reduce the gradient of row i over its (b,k) occurrences (item-major list built
//     by P0-P2), then apply AdamW to the row in the same wave (dense semantics: untouched rows get
//     the zero-gradient update).  One wave per item.  APPLY=false writes the dense gradient instead.
// =============================================================================================
template <int V, int NCH, bool FULL, int R, bool APPLY>
__global__ __launch_bounds__(256) void k_item_update(const float* __restrict__ Uw, float* __restrict__ Iw,
                                                     float* __restrict__ Ib, float* __restrict__ mI,
                                                     float* __restrict__ vI, float* __restrict__ mIb,
                                                     float* __restrict__ vIb, const int* __restrict__ u32,
                                                     const float* __restrict__ g_s, const int* __restrict__ perm,
                                                     const int* __restrict__ offsets, int* __restrict__ counts,
                                                     int n_items, int K, int D, hsk_adamw_consts c,
                                                     float* __restrict__ gI_out, float* __restrict__ gIb_out) {
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int i = blockIdx.x * 4 + wave;
  if (i >= n_items) return;
  using Row = hsk_row<V, NCH>;

  const int beg = hsk_uniform_i(offsets[i]);
  const int end = hsk_uniform_i(offsets[i + 1]);
  Row acc;
  hsk_row_zero(acc);
  float gb_lane = 0.f;

  for (int c0 = beg; c0 < end; c0 += 64) {
    const int nr = min(64, end - c0);
    int myu = 0;
    float myg = 0.f;
    if (lane < nr) {
      const int e = perm[c0 + lane];
      myg = g_s[e];
      myu = u32[e / K];
    }
    gb_lane += myg;

    Row bufA[R], bufB[R];
    auto prefetch = [&](Row(&buf)[R], int j) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (j + r < nr)
          hsk_row_load<V, NCH, FULL>(buf[r], Uw + (long long)hsk_readlane_i(myu, j + r) * D, lane, D);
    };
    auto process = [&](Row(&buf)[R], int j) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (j + r < nr) hsk_row_axpy(acc, hsk_readlane_f(myg, j + r), buf[r]);
    };
    prefetch(bufA, 0);
    for (int j = 0; j < nr; j += 2 * R) {
      prefetch(bufB, j + R);
      process(bufA, j);
      prefetch(bufA, j + 2 * R);
      process(bufB, j + R);
    }
  }
  const float gbias = hsk_wave_sum(gb_lane);

  if (APPLY) {
    float* prow = Iw + (long long)i * D;
    float* mrow = mI + (long long)i * D;
    float* vrow = vI + (long long)i * D;
    Row p, m, v;
    hsk_row_load<V, NCH, FULL>(p, prow, lane, D);
    hsk_row_load<V, NCH, FULL>(m, mrow, lane, D);
    hsk_row_load<V, NCH, FULL>(v, vrow, lane, D);
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
      for (int q = 0; q < V; ++q) hsk_adamw_update(p.c[cc].v[q], m.c[cc].v[q], v.c[cc].v[q], acc.c[cc].v[q], c);
    hsk_row_store<V, NCH, FULL>(p, prow, lane, D);
    hsk_row_store<V, NCH, FULL>(m, mrow, lane, D);
    hsk_row_store<V, NCH, FULL>(v, vrow, lane, D);
    if (lane == 0) {
      if (Ib) {
        float pb = Ib[i], mb = mIb[i], vb = vIb[i];
        hsk_adamw_update(pb, mb, vb, gbias, c);
        Ib[i] = pb;
        mIb[i] = mb;
        vIb[i] = vb;
      }
      counts[i] = 0;  // histogram consumed; ready for the next step
    }
  } else {
    hsk_row_store<V, NCH, FULL>(acc, gI_out + (long long)i * D, lane, D);
    if (lane == 0) {
      if (gIb_out) gIb_out[i] = gbias;
      counts[i] = 0;
    }
  }
}

// =============================================================================================
// K3: user rows.  Row gradient = sum of dUb[b] over the batch entries with u32[b] == row, added in
//     ascending b (deterministic).  owner[row] = min b, cnt[row] = multiplicity (from P0).
//     MODE 0: dense AdamW sweep over all rows (one wave per table row)
//     MODE 1: dense gradient output (compat backward)
// =============================================================================================
template <int V, int NCH, bool FULL>
__device__ __forceinline__ void hsk_user_grad(hsk_row<V, NCH>& acc, float& gbias_unused, int row, int b0, int c,
                                              const float* __restrict__ dUb, const int* __restrict__ u32, int B,
                                              int D, int lane) {
  hsk_row_load<V, NCH, FULL>(acc, dUb + (long long)b0 * D, lane, D);
  if (c > 1) {
    for (int c0 = (b0 / 64) * 64; c0 < B; c0 += 64) {
      const int bb = c0 + lane;
      const bool match = (bb < B) && (bb > b0) && (u32[bb] == row);
      unsigned long long mask = __ballot(match);
      while (mask) {
        const int j = __builtin_ctzll(mask);
        mask &= mask - 1;
        hsk_row<V, NCH> t;
        hsk_row_load<V, NCH, FULL>(t, dUb + (long long)(c0 + j) * D, lane, D);
        hsk_row_add(acc, t);
      }
    }
  }
}

template <int V, int NCH, bool FULL, int MODE>
__global__ __launch_bounds__(256) void k_user_update(float* __restrict__ Uw, float* __restrict__ mU,
                                                     float* __restrict__ vU, float* __restrict__ Ub,
                                                     float* __restrict__ mUb, float* __restrict__ vUb,
                                                     const float* __restrict__ dUb, const int* __restrict__ u32,
                                                     int* __restrict__ owner, int* __restrict__ cnt, int n_users,
                                                     int B, int D, hsk_adamw_consts c,
                                                     float* __restrict__ gU_out, float* __restrict__ gUb_out) {
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int row = blockIdx.x * 4 + wave;
  if (row >= n_users) return;
  using Row = hsk_row<V, NCH>;
  const int n = hsk_uniform_i(cnt[row]);
  Row g;
  float dummy = 0.f;
  if (n > 0) {
    const int b0 = hsk_uniform_i(owner[row]);
    hsk_user_grad<V, NCH, FULL>(g, dummy, row, b0, n, dUb, u32, B, D, lane);
    if (lane == 0) {
      owner[row] = HSK_OWNER_NONE;
      cnt[row] = 0;
    }
  } else {
    hsk_row_zero(g);
  }
  if (MODE == 0) {
    float* prow = Uw + (long long)row * D;
    float* mrow = mU + (long long)row * D;
    float* vrow = vU + (long long)row * D;
    Row p, m, v;
    hsk_row_load<V, NCH, FULL>(p, prow, lane, D);
    hsk_row_load<V, NCH, FULL>(m, mrow, lane, D);
    hsk_row_load<V, NCH, FULL>(v, vrow, lane, D);
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
      for (int q = 0; q < V; ++q) hsk_adamw_update(p.c[cc].v[q], m.c[cc].v[q], v.c[cc].v[q], g.c[cc].v[q], c);
    hsk_row_store<V, NCH, FULL>(p, prow, lane, D);
    hsk_row_store<V, NCH, FULL>(m, mrow, lane, D);
    hsk_row_store<V, NCH, FULL>(v, vrow, lane, D);
    if (Ub && lane == 0) {
      // d loss / d user_bias is identically 0 under BPR (it cancels in s_pos - s_neg)
      float pb = Ub[row], mb = mUb[row], vb = vUb[row];
      hsk_adamw_update(pb, mb, vb, 0.f, c);
      Ub[row] = pb;
      mUb[row] = mb;
      vUb[row] = vb;
    }
  } else {
    hsk_row_store<V, NCH, FULL>(g, gU_out + (long long)row * D, lane, D);
  }
}

// tail: deterministic fp64 reduction of the per-positive loss sums; global-bias zero-grad AdamW step
__global__ __launch_bounds__(1024) void k_finish_step(const double* __restrict__ loss_b, int B, double inv_bn,
                                                      double* __restrict__ loss_out, float* gb, float* mgb,
                                                      float* vgb, hsk_adamw_consts c) {
  __shared__ double red[1024];
  const int t = threadIdx.x;
  double s = 0.0;
  for (int i = t; i < B; i += 1024) s += loss_b[i];
  red[t] = s;
  __syncthreads();
  for (int off = 512; off >= 1; off >>= 1) {
    if (t < off) red[t] += red[t + off];
    __syncthreads();
  }
  if (t == 0) {
    const double loss = red[0] * inv_bn;
    if (loss_out) {
      loss_out[0] = loss;
      loss_out[1] += loss;
    }
    if (gb) {
      float p = gb[0], m = mgb[0], v = vgb[0];
      hsk_adamw_update(p, m, v, 0.f, c);
      gb[0] = p;
      mgb[0] = m;
      vgb[0] = v;
    }
  }
}

__global__ __launch_bounds__(256) void k_fill_i32(int* p, long long n, int v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ __launch_bounds__(256) void k_widen_batch(const int* __restrict__ u32, const int* __restrict__ it32,
                                                     long long B, long long total, int64_t* __restrict__ u_out,
                                                     int64_t* __restrict__ i_out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) i_out[e] = it32[e];
  if (e < B) u_out[e] = u32[e];
}

// =============================================================================================
// workspace carving
// =============================================================================================
struct hsk_ws {
  int* u32;
  int* it32;
  float* g_s;
  int* perm;
  int* counts;
  int* offsets;
  int* cursor;
  int* owner;
  int* cnt;
  int* last_step;
  float* dUb;
  double* loss_b;
  int64_t total;
};

static hsk_ws hsk_carve(void* base, int64_t n_users, int64_t n_items, int64_t dim, int64_t max_batch,
                        int64_t max_cols) {
  hsk_ws w;
  char* p = (char*)base;
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    char* r = p ? p + off : nullptr;
    off += hsk_align_up(bytes, 256);
    return r;
  };
  const int64_t ent = max_batch * max_cols;
  w.u32 = (int*)take(max_batch * 4);
  w.it32 = (int*)take(ent * 4);
  w.g_s = (float*)take(ent * 4);
  w.perm = (int*)take(ent * 4);
  w.counts = (int*)take(n_items * 4);
  w.offsets = (int*)take((n_items + 1) * 4);
  w.cursor = (int*)take(n_items * 4);
  w.owner = (int*)take(n_users * 4);
  w.cnt = (int*)take(n_users * 4);
  w.last_step = (int*)take(n_users * 4);
  w.dUb = (float*)take(max_batch * dim * 4);
  w.loss_b = (double*)take(max_batch * 8);
  w.total = off;
  return w;
}

extern "C" int64_t hsk_bprmf_workspace_bytes(int64_t n_users, int64_t n_items, int64_t dim, int64_t max_batch,
                                             int64_t max_cols) {
  if (n_users <= 0 || n_items <= 0 || dim <= 0 || max_batch <= 0 || max_cols <= 1) return -1;
  return hsk_carve(nullptr, n_users, n_items, dim, max_batch, max_cols).total;
}

static int hsk_check_state(const hsk_bprmf_state* st) {
  HSK_REQUIRE(st != nullptr, HSK_ERR_INVALID, "state is NULL");
  HSK_REQUIRE(st->user_emb && st->item_emb && st->m_user_emb && st->v_user_emb && st->m_item_emb && st->v_item_emb,
              HSK_ERR_INVALID, "embedding / moment pointers must not be NULL");
  HSK_REQUIRE(!st->item_bias || (st->m_item_bias && st->v_item_bias), HSK_ERR_INVALID, "item_bias moments missing");
  HSK_REQUIRE(!st->user_bias || (st->m_user_bias && st->v_user_bias), HSK_ERR_INVALID, "user_bias moments missing");
  HSK_REQUIRE(!st->global_bias || (st->m_global_bias && st->v_global_bias), HSK_ERR_INVALID,
              "global_bias moments missing");
  HSK_REQUIRE(st->n_users > 0 && st->n_items > 0 && st->dim > 0, HSK_ERR_INVALID, "bad table shape");
  HSK_REQUIRE(st->n_users < 0x7fffffff && st->n_items < 0x7fffffff, HSK_ERR_UNSUPPORTED, "tables too large for int32 ids");
  HSK_REQUIRE(st->workspace != nullptr, HSK_ERR_INVALID, "workspace is NULL");
  const int64_t need = hsk_bprmf_workspace_bytes(st->n_users, st->n_items, st->dim, st->max_batch, st->max_cols);
  HSK_REQUIRE(need > 0 && st->workspace_bytes >= need, HSK_ERR_INVALID, "workspace too small: %lld < %lld",
              (long long)st->workspace_bytes, (long long)need);
  HSK_REQUIRE(((uintptr_t)st->workspace & 255) == 0, HSK_ERR_INVALID, "workspace must be 256-byte aligned");
  const int V = (st->dim % 4 == 0) ? 4 : (st->dim % 2 == 0) ? 2 : 1;
  HSK_REQUIRE((((uintptr_t)st->user_emb | (uintptr_t)st->item_emb | (uintptr_t)st->m_user_emb |
                (uintptr_t)st->v_user_emb | (uintptr_t)st->m_item_emb | (uintptr_t)st->v_item_emb) &
               (uintptr_t)(4 * V - 1)) == 0,
              HSK_ERR_INVALID, "tables must be %d-byte aligned", 4 * V);
  HSK_REQUIRE(st->lazy_users == 0, HSK_ERR_UNSUPPORTED, "lazy_users not available in this build");
  return HSK_OK;
}

extern "C" int hsk_bprmf_init_workspace(const hsk_bprmf_state* st, hsk_stream_t stream_) {
  int rc = hsk_check_state(st);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  hsk_ws w = hsk_carve(st->workspace, st->n_users, st->n_items, st->dim, st->max_batch, st->max_cols);
  HSK_HIP(hipMemsetAsync(w.counts, 0, st->n_items * 4, stream));
  HSK_HIP(hipMemsetAsync(w.cnt, 0, st->n_users * 4, stream));
  HSK_HIP(hipMemsetAsync(w.last_step, 0, st->n_users * 4, stream));
  k_fill_i32<<<(unsigned)hsk_ceil_div(st->n_users, 256), 256, 0, stream>>>(w.owner, st->n_users, HSK_OWNER_NONE);
  HSK_LAUNCH_CHECK();
  if (st->loss_out) HSK_HIP(hipMemsetAsync(st->loss_out, 0, 2 * sizeof(double), stream));
  if (st->status) HSK_HIP(hipMemsetAsync(st->status, 0, sizeof(int32_t), stream));
  return HSK_OK;
}

// ---------------------------------------------------------------------------------------------
// per-stage HIP-event timing (host-side recorder; events are recorded on the kernels' own stream)
// ---------------------------------------------------------------------------------------------
#include <vector>
struct hsk_timing {
  std::vector<hipEvent_t> beg[HSK_STAGE_COUNT], end[HSK_STAGE_COUNT];
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) {
      hipEvent_t e = pool.back();
      pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
};

extern "C" void* hsk_timing_create(void) { return new hsk_timing(); }

extern "C" void hsk_timing_destroy(void* t_) {
  hsk_timing* t = (hsk_timing*)t_;
  if (!t) return;
  for (int s = 0; s < HSK_STAGE_COUNT; ++s) {
    for (hipEvent_t e : t->beg[s]) (void)hipEventDestroy(e);
    for (hipEvent_t e : t->end[s]) (void)hipEventDestroy(e);
  }
  for (hipEvent_t e : t->pool) (void)hipEventDestroy(e);
  delete t;
}

extern "C" int hsk_timing_collect(void* t_, double* ms_sum, int64_t* count) {
  hsk_timing* t = (hsk_timing*)t_;
  HSK_REQUIRE(t && ms_sum && count, HSK_ERR_INVALID, "NULL argument");
  for (int s = 0; s < HSK_STAGE_COUNT; ++s) {
    for (size_t i = 0; i < t->end[s].size(); ++i) {
      HSK_HIP(hipEventSynchronize(t->end[s][i]));
      float ms = 0.f;
      HSK_HIP(hipEventElapsedTime(&ms, t->beg[s][i], t->end[s][i]));
      ms_sum[s] += (double)ms;
      count[s] += 1;
      t->pool.push_back(t->beg[s][i]);
      t->pool.push_back(t->end[s][i]);
    }
    t->beg[s].clear();
    t->end[s].clear();
  }
  return HSK_OK;
}

static inline void hsk_stage_mark(const hsk_bprmf_state* st, int stage, bool begin, hipStream_t stream) {
  hsk_timing* t = (hsk_timing*)st->timing;
  if (!t || !((st->timing_mask >> stage) & 1)) return;
  hipEvent_t e = t->get();
  if (!e) return;
  (void)hipEventRecord(e, stream);
  (begin ? t->beg[stage] : t->end[stage]).push_back(e);
}

#define HSK_STAGE(stage, ...)                      \
  do {                                             \
    hsk_stage_mark(st, (stage), true, stream);     \
    __VA_ARGS__;                                   \
    hsk_stage_mark(st, (stage), false, stream);    \
  } while (0)

// stages shared by the external-batch and device-sampled steps (after P0 filled u32/it32/counts/owner/cnt)
static int hsk_run_step(hsk_bprmf_state* st, const hsk_ws& w, int64_t B, int64_t K, hipStream_t stream) {
  const int64_t total = B * K;
  const int I = (int)st->n_items, U = (int)st->n_users, D = (int)st->dim;
  st->step += 1;
  const hsk_adamw_consts c = hsk_make_adamw_consts(st->lr, st->beta1, st->beta2, st->eps, st->wd, st->step);
  const double inv_bn_d = 1.0 / ((double)B * (double)(K - 1));
  const float inv_bn = (float)inv_bn_d;

  HSK_STAGE(HSK_STAGE_SCAN, k_scan_counts<<<1, 1024, 0, stream>>>(w.counts, I, w.offsets, w.cursor));
  HSK_LAUNCH_CHECK();
  HSK_STAGE(HSK_STAGE_SCATTER,
            k_scatter_perm<<<(unsigned)hsk_ceil_div(total, 256), 256, 0, stream>>>(w.it32, total, w.cursor, w.perm));
  HSK_LAUNCH_CHECK();

  int rc = hsk_dispatch_dim(D, [&](auto v_, auto n_, auto f_) {
    constexpr int V = decltype(v_)::value;
    constexpr int NCH = decltype(n_)::value;
    constexpr bool FULL = decltype(f_)::value;
    constexpr int R = (V * NCH >= 16) ? 2 : 4;
    HSK_STAGE(HSK_STAGE_FWD, (k_fwd_ugrad<V, NCH, FULL, R><<<(unsigned)hsk_ceil_div(B, 4), 256, 0, stream>>>(
                                 st->user_emb, st->item_emb, st->item_bias, w.u32, w.it32, (int)B, (int)K, D, inv_bn,
                                 w.g_s, w.dUb, w.loss_b)));
    HSK_STAGE(HSK_STAGE_ITEM, (k_item_update<V, NCH, FULL, R, true><<<(unsigned)hsk_ceil_div(I, 4), 256, 0, stream>>>(
                                  st->user_emb, st->item_emb, st->item_bias, st->m_item_emb, st->v_item_emb,
                                  st->m_item_bias, st->v_item_bias, w.u32, w.g_s, w.perm, w.offsets, w.counts, I, (int)K,
                                  D, c, nullptr, nullptr)));
    HSK_STAGE(HSK_STAGE_USER, (k_user_update<V, NCH, FULL, 0><<<(unsigned)hsk_ceil_div(U, 4), 256, 0, stream>>>(
                                  st->user_emb, st->m_user_emb, st->v_user_emb, st->user_bias, st->m_user_bias,
                                  st->v_user_bias, w.dUb, w.u32, w.owner, w.cnt, U, (int)B, D, c, nullptr, nullptr)));
    return HSK_OK;
  });
  if (rc) return rc;
  HSK_LAUNCH_CHECK();
  HSK_STAGE(HSK_STAGE_FINISH, k_finish_step<<<1, 1024, 0, stream>>>(w.loss_b, (int)B, inv_bn_d, st->loss_out,
                                                                     st->global_bias, st->m_global_bias,
                                                                     st->v_global_bias, c));
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

extern "C" int hsk_bprmf_train_step(hsk_bprmf_state* st, const int64_t* u_idx, const int64_t* i_idx, int64_t batch,
                                    int64_t n_cols, hsk_stream_t stream_) {
  int rc = hsk_check_state(st);
  if (rc) return rc;
  HSK_REQUIRE(u_idx && i_idx, HSK_ERR_INVALID, "u_idx / i_idx must not be NULL");
  HSK_REQUIRE(batch > 0 && batch <= st->max_batch, HSK_ERR_INVALID, "batch %lld outside (0, %lld]", (long long)batch,
              (long long)st->max_batch);
  HSK_REQUIRE(n_cols >= 2 && n_cols <= st->max_cols, HSK_ERR_INVALID, "n_cols %lld outside [2, %lld]",
              (long long)n_cols, (long long)st->max_cols);
  hipStream_t stream = (hipStream_t)stream_;
  hsk_ws w = hsk_carve(st->workspace, st->n_users, st->n_items, st->dim, st->max_batch, st->max_cols);
  const int64_t total = batch * n_cols;
  HSK_STAGE(HSK_STAGE_PREP, k_prep_external<<<(unsigned)hsk_ceil_div(total, 256), 256, 0, stream>>>(
                                u_idx, i_idx, (int)batch, (int)n_cols, (int)st->n_users, (int)st->n_items, w.u32,
                                w.it32, w.counts, w.owner, w.cnt, st->status));
  HSK_LAUNCH_CHECK();
  return hsk_run_step(st, w, batch, n_cols, stream);
}

extern "C" int hsk_bprmf_train_step_sampled(hsk_bprmf_state* st, const int64_t* order, int64_t start, int64_t batch,
                                            int64_t n_neg, hsk_stream_t stream_) {
  int rc = hsk_check_state(st);
  if (rc) return rc;
  HSK_REQUIRE(st->csr_indptr && st->csr_indices && st->coo_user && st->coo_item, HSK_ERR_INVALID,
              "CSR/COO of the training interactions missing from the state");
  HSK_REQUIRE(batch > 0 && batch <= st->max_batch, HSK_ERR_INVALID, "batch %lld outside (0, %lld]", (long long)batch,
              (long long)st->max_batch);
  HSK_REQUIRE(n_neg >= 1 && n_neg + 1 <= st->max_cols, HSK_ERR_INVALID, "n_neg %lld outside [1, %lld]",
              (long long)n_neg, (long long)st->max_cols - 1);
  HSK_REQUIRE(start >= 0 && start + batch <= st->nnz, HSK_ERR_INVALID, "interaction range [%lld, %lld) outside nnz %lld",
              (long long)start, (long long)(start + batch), (long long)st->nnz);
  hipStream_t stream = (hipStream_t)stream_;
  hsk_ws w = hsk_carve(st->workspace, st->n_users, st->n_items, st->dim, st->max_batch, st->max_cols);
  // the RNG stream id is the index of the step about to be taken: every step draws fresh negatives
  HSK_STAGE(HSK_STAGE_PREP, k_prep_sample<<<(unsigned)hsk_ceil_div(batch, 4), 256, 0, stream>>>(
                                st->coo_user, st->coo_item, order, start, (int)batch, (int)n_neg, st->csr_indptr,
                                st->csr_indices, (int)st->n_items, st->seed, (uint64_t)st->step, w.u32, w.it32,
                                w.counts, w.owner, w.cnt, st->status));
  HSK_LAUNCH_CHECK();
  return hsk_run_step(st, w, batch, n_neg + 1, stream);
}

extern "C" int hsk_bprmf_flush(hsk_bprmf_state* st, hsk_stream_t) {
  int rc = hsk_check_state(st);
  if (rc) return rc;
  return HSK_OK;  // dense user updates: nothing is pending
}

extern "C" int hsk_bprmf_last_batch(const hsk_bprmf_state* st, int64_t batch, int64_t n_cols, int64_t* u_out,
                                    int64_t* i_out, hsk_stream_t stream_) {
  int rc = hsk_check_state(st);
  if (rc) return rc;
  HSK_REQUIRE(u_out && i_out, HSK_ERR_INVALID, "output pointers must not be NULL");
  HSK_REQUIRE(batch > 0 && batch <= st->max_batch && n_cols >= 2 && n_cols <= st->max_cols, HSK_ERR_INVALID,
              "bad batch shape");
  hsk_ws w = hsk_carve(st->workspace, st->n_users, st->n_items, st->dim, st->max_batch, st->max_cols);
  const int64_t total = batch * n_cols;
  k_widen_batch<<<(unsigned)hsk_ceil_div(total, 256), 256, 0, (hipStream_t)stream_>>>(w.u32, w.it32, batch, total,
                                                                                      u_out, i_out);
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

extern "C" int hsk_sample_negatives_uniform(const int64_t* csr_indptr, const int32_t* csr_indices, int64_t n_users,
                                            int64_t n_items, const int64_t* u_idx, int64_t batch, int64_t n_neg,
                                            uint64_t seed, uint64_t stream_id, int64_t* neg_out, int32_t* status,
                                            hsk_stream_t stream_) {
  HSK_REQUIRE(csr_indptr && csr_indices && u_idx && neg_out, HSK_ERR_INVALID, "NULL pointer argument");
  HSK_REQUIRE(n_users > 0 && n_items > 0 && n_items < 0x7fffffff && n_users < 0x7fffffff, HSK_ERR_INVALID,
              "bad n_users / n_items");
  HSK_REQUIRE(batch >= 0 && n_neg >= 1, HSK_ERR_INVALID, "bad batch / n_neg");
  if (batch == 0) return HSK_OK;
  k_sample_negatives<<<(unsigned)hsk_ceil_div(batch, 4), 256, 0, (hipStream_t)stream_>>>(
      csr_indptr, csr_indices, (int)n_users, (int)n_items, u_idx, (int)batch, (int)n_neg, seed, stream_id, neg_out,
      status);
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

// =============================================================================================
// Un-fused operators for the autograd path
// =============================================================================================
#define HSK_SCORE_COLS 128  // columns handled by one wave

template <int V, int NCH, bool FULL, int R>
__global__ __launch_bounds__(256) void k_scores(const float* __restrict__ Uw, const float* __restrict__ Iw,
                                                const float* __restrict__ Ib, const float* __restrict__ Ub,
                                                const float* __restrict__ gb, int n_users, int n_items, int D,
                                                const int64_t* __restrict__ u_idx, const int64_t* __restrict__ i_idx,
                                                int B, long long K, float* __restrict__ logits, int32_t* status) {
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const long long k0 = ((long long)blockIdx.x * 4 + wave) * HSK_SCORE_COLS;
  if (k0 >= K) return;
  using Row = hsk_row<V, NCH>;
  const int u = hsk_uniform_i(hsk_clamp_index(u_idx[b], n_users, status));
  Row ur;
  hsk_row_load<V, NCH, FULL>(ur, Uw + (long long)u * D, lane, D);
  const int ncols = (int)min((long long)HSK_SCORE_COLS, K - k0);
  for (int kc = 0; kc < ncols; kc += 64) {
    const int nr = min(64, ncols - kc);
    int myidx = 0;
    if (lane < nr) myidx = hsk_clamp_index(i_idx[(long long)b * K + k0 + kc + lane], n_items, status);
    const float mybias = Ib ? Ib[myidx] : 0.f;
    float sv = 0.f;
    for (int j = 0; j < nr; j += R) {
      Row buf[R];
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (j + r < nr) hsk_row_load<V, NCH, FULL>(buf[r], Iw + (long long)hsk_readlane_i(myidx, j + r) * D, lane, D);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (j + r < nr) {
          const float s = hsk_wave_sum(hsk_row_dot_partial(ur, buf[r]));
          sv = (lane == j + r) ? s : sv;
        }
    }
    // reference order of the bias adds: out += u_bias; out += i_bias; out += global_bias
    if (lane < nr) {
      float o = sv;
      if (Ub) o += Ub[u];
      if (Ib) o += mybias;
      if (gb) o += gb[0];
      logits[(long long)b * K + k0 + kc + lane] = o;
    }
  }
}

extern "C" int hsk_mf_scores(const float* user_emb, const float* item_emb, const float* item_bias,
                             const float* user_bias, const float* global_bias, int64_t n_users, int64_t n_items,
                             int64_t dim, const int64_t* u_idx, const int64_t* i_idx, int64_t batch, int64_t n_cols,
                             float* logits, int32_t* status, hsk_stream_t stream_) {
  HSK_REQUIRE(user_emb && item_emb && u_idx && i_idx && logits, HSK_ERR_INVALID, "NULL pointer argument");
  HSK_REQUIRE(n_users > 0 && n_items > 0 && dim > 0 && batch >= 0 && n_cols >= 0, HSK_ERR_INVALID, "bad sizes");
  HSK_REQUIRE(batch <= 65535, HSK_ERR_UNSUPPORTED, "batch %lld > 65535 rows per call", (long long)batch);
  if (batch == 0 || n_cols == 0) return HSK_OK;
  hipStream_t stream = (hipStream_t)stream_;
  int rc = hsk_dispatch_dim(dim, [&](auto v_, auto n_, auto f_) {
    constexpr int V = decltype(v_)::value;
    constexpr int NCH = decltype(n_)::value;
    constexpr bool FULL = decltype(f_)::value;
    constexpr int R = (V * NCH >= 16) ? 2 : 4;
    HSK_REQUIRE((((uintptr_t)user_emb | (uintptr_t)item_emb) & (uintptr_t)(4 * V - 1)) == 0, HSK_ERR_INVALID,
                "tables must be %d-byte aligned", 4 * V);
    dim3 grid((unsigned)hsk_ceil_div(n_cols, (int64_t)HSK_SCORE_COLS * 4), (unsigned)batch);
    k_scores<V, NCH, FULL, R><<<grid, 256, 0, stream>>>(user_emb, item_emb, item_bias, user_bias, global_bias,
                                                         (int)n_users, (int)n_items, (int)dim, u_idx, i_idx, (int)batch,
                                                         (long long)n_cols, logits, status);
    return HSK_OK;
  });
  if (rc) return rc;
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

// BPR loss + gradient wrt logits, one wave per row
__global__ __launch_bounds__(256) void k_bpr_loss_grad(const float* __restrict__ logits, int B, long long K,
                                                       float inv_bn, float* __restrict__ grad, double* __restrict__ ws) {
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const float* row = logits + (long long)b * K;
  const float s0 = row[0];
  double lsum = 0.0;
  float gsum = 0.f;
  for (long long k = 1 + lane; k < K; k += 64) {
    const float x = s0 - row[k];
    lsum += (double)hsk_softplus(-x);
    const float g = inv_bn / (1.f + expf(x));
    gsum += g;
    if (grad) grad[(long long)b * K + k] = g;
  }
  const double l = hsk_wave_sum_f64(lsum);
  const float gs = hsk_wave_sum(gsum);
  if (lane == 0) {
    ws[b] = l;
    if (grad) grad[(long long)b * K] = -gs;
  }
}

// deterministic fp64 tree reduction: out[0] = scale * sum(x[0..n))
__global__ __launch_bounds__(1024) void k_loss_mean(const double* __restrict__ x, int n, double scale,
                                                    double* __restrict__ out) {
  __shared__ double red[1024];
  const int t = threadIdx.x;
  double s = 0.0;
  for (int i = t; i < n; i += 1024) s += x[i];
  red[t] = s;
  __syncthreads();
  for (int off = 512; off >= 1; off >>= 1) {
    if (t < off) red[t] += red[t + off];
    __syncthreads();
  }
  if (t == 0) out[0] = red[0] * scale;
}

extern "C" int hsk_bpr_loss_grad(const float* logits, int64_t batch, int64_t n_cols, double* loss, float* grad_logits,
                                 double* ws, hsk_stream_t stream_) {
  HSK_REQUIRE(logits && loss && ws, HSK_ERR_INVALID, "NULL pointer argument");
  HSK_REQUIRE(batch > 0 && n_cols >= 2, HSK_ERR_INVALID, "need batch > 0 and at least one negative column");
  HSK_REQUIRE(batch < 0x7fffffff, HSK_ERR_UNSUPPORTED, "batch too large");
  hipStream_t stream = (hipStream_t)stream_;
  const double inv_bn = 1.0 / ((double)batch * (double)(n_cols - 1));
  k_bpr_loss_grad<<<(unsigned)hsk_ceil_div(batch, 4), 256, 0, stream>>>(logits, (int)batch, (long long)n_cols,
                                                                        (float)inv_bn, grad_logits, ws);
  HSK_LAUNCH_CHECK();
  k_loss_mean<<<1, 1024, 0, stream>>>(ws, (int)batch, inv_bn, loss);
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

// Dense gradients for the autograd path: float atomics into zero-filled [U,D] / [I,D] buffers.
// One wave per (b, chunk of HSK_SCORE_COLS columns).  (The fused step uses the atomic-free
// item-major reduction instead; this operator only exists so an unmodified Trainer.fit +
// torch.optim works on top of the HIP forward.)
template <int V, int NCH, bool FULL>
__global__ __launch_bounds__(256) void k_backward_dense(const float* __restrict__ Uw, const float* __restrict__ Iw,
                                                        int n_users, int n_items, int D,
                                                        const int64_t* __restrict__ u_idx,
                                                        const int64_t* __restrict__ i_idx, int B, long long K,
                                                        const float* __restrict__ gl, float* __restrict__ gU,
                                                        float* __restrict__ gI, float* __restrict__ gIb,
                                                        float* __restrict__ gUb, float* __restrict__ ggb,
                                                        int32_t* status) {
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const long long k0 = ((long long)blockIdx.x * 4 + wave) * HSK_SCORE_COLS;
  if (k0 >= K) return;
  using Row = hsk_row<V, NCH>;
  const int u = hsk_uniform_i(hsk_clamp_index(u_idx[b], n_users, status));
  Row ur, acc;
  hsk_row_load<V, NCH, FULL>(ur, Uw + (long long)u * D, lane, D);
  hsk_row_zero(acc);
  float gsum_lane = 0.f;
  const int ncols = (int)min((long long)HSK_SCORE_COLS, K - k0);
  for (int kc = 0; kc < ncols; kc += 64) {
    const int nr = min(64, ncols - kc);
    int myidx = 0;
    float myg = 0.f;
    if (lane < nr) {
      myidx = hsk_clamp_index(i_idx[(long long)b * K + k0 + kc + lane], n_items, status);
      myg = gl[(long long)b * K + k0 + kc + lane];
      if (gIb) atomicAdd(&gIb[myidx], myg);
    }
    gsum_lane += myg;
    for (int j = 0; j < nr; ++j) {
      const int it = hsk_readlane_i(myidx, j);
      const float g = hsk_readlane_f(myg, j);
      Row r;
      hsk_row_load<V, NCH, FULL>(r, Iw + (long long)it * D, lane, D);
      hsk_row_axpy(acc, g, r);
      if (gI) {
        float* dst = gI + (long long)it * D;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int off = (c * 64 + lane) * V;
          if (FULL || off < D) {
#pragma unroll
            for (int q = 0; q < V; ++q) atomicAdd(dst + off + q, g * ur.c[c].v[q]);
          }
        }
      }
    }
  }
  if (gU) {
    float* dst = gU + (long long)u * D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int off = (c * 64 + lane) * V;
      if (FULL || off < D) {
#pragma unroll
        for (int q = 0; q < V; ++q) atomicAdd(dst + off + q, acc.c[c].v[q]);
      }
    }
  }
  const float gs = hsk_wave_sum(gsum_lane);
  if (lane == 0) {
    if (gUb) atomicAdd(&gUb[u], gs);
    if (ggb) atomicAdd(ggb, gs);
  }
}

extern "C" int hsk_mf_backward(const float* user_emb, const float* item_emb, int64_t n_users, int64_t n_items,
                               int64_t dim, const int64_t* u_idx, const int64_t* i_idx, int64_t batch, int64_t n_cols,
                               const float* grad_logits, float* g_user_emb, float* g_item_emb, float* g_item_bias,
                               float* g_user_bias, float* g_global_bias, int32_t* status, hsk_stream_t stream_) {
  HSK_REQUIRE(user_emb && item_emb && u_idx && i_idx && grad_logits, HSK_ERR_INVALID, "NULL pointer argument");
  HSK_REQUIRE(n_users > 0 && n_items > 0 && dim > 0 && batch >= 0 && n_cols >= 0, HSK_ERR_INVALID, "bad sizes");
  HSK_REQUIRE(batch <= 65535, HSK_ERR_UNSUPPORTED, "batch %lld > 65535 rows per call", (long long)batch);
  hipStream_t stream = (hipStream_t)stream_;
  if (g_user_emb) HSK_HIP(hipMemsetAsync(g_user_emb, 0, (size_t)n_users * dim * 4, stream));
  if (g_item_emb) HSK_HIP(hipMemsetAsync(g_item_emb, 0, (size_t)n_items * dim * 4, stream));
  if (g_item_bias) HSK_HIP(hipMemsetAsync(g_item_bias, 0, (size_t)n_items * 4, stream));
  if (g_user_bias) HSK_HIP(hipMemsetAsync(g_user_bias, 0, (size_t)n_users * 4, stream));
  if (g_global_bias) HSK_HIP(hipMemsetAsync(g_global_bias, 0, 4, stream));
  if (batch == 0 || n_cols == 0) return HSK_OK;
  int rc = hsk_dispatch_dim(dim, [&](auto v_, auto n_, auto f_) {
    constexpr int V = decltype(v_)::value;
    constexpr int NCH = decltype(n_)::value;
    constexpr bool FULL = decltype(f_)::value;
    HSK_REQUIRE((((uintptr_t)user_emb | (uintptr_t)item_emb) & (uintptr_t)(4 * V - 1)) == 0, HSK_ERR_INVALID,
                "tables must be %d-byte aligned", 4 * V);
    dim3 grid((unsigned)hsk_ceil_div(n_cols, (int64_t)HSK_SCORE_COLS * 4), (unsigned)batch);
    k_backward_dense<V, NCH, FULL><<<grid, 256, 0, stream>>>(user_emb, item_emb, (int)n_users, (int)n_items, (int)dim,
                                                             u_idx, i_idx, (int)batch, (long long)n_cols, grad_logits,
                                                             g_user_emb, g_item_emb, g_item_bias, g_user_bias,
                                                             g_global_bias, status);
    return HSK_OK;
  });
  if (rc) return rc;
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

// flat dense AdamW (float4 main body + scalar tail); g == NULL means zero gradient
__global__ __launch_bounds__(256) void k_adamw_dense(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v, long long n,
                                                     hsk_adamw_consts c) {
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    hsk_vec<4> pp = hsk_ldg<4>(p + 4 * i), mm = hsk_ldg<4>(m + 4 * i), vv = hsk_ldg<4>(v + 4 * i);
    hsk_vec<4> gg = g ? hsk_ldg<4>(g + 4 * i) : hsk_zero<4>();
#pragma unroll
    for (int q = 0; q < 4; ++q) hsk_adamw_update(pp.v[q], mm.v[q], vv.v[q], gg.v[q], c);
    hsk_stg<4>(p + 4 * i, pp);
    hsk_stg<4>(m + 4 * i, mm);
    hsk_stg<4>(v + 4 * i, vv);
  }
  const long long t = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    float pp = p[t], mm = m[t], vv = v[t];
    hsk_adamw_update(pp, mm, vv, g ? g[t] : 0.f, c);
    p[t] = pp;
    m[t] = mm;
    v[t] = vv;
  }
}

extern "C" int hsk_adamw_dense(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                               double beta2, double eps, double wd, int64_t step, hsk_stream_t stream_) {
  HSK_REQUIRE(p && m && v, HSK_ERR_INVALID, "NULL pointer argument");
  HSK_REQUIRE(n >= 0 && step >= 1, HSK_ERR_INVALID, "need n >= 0 and step >= 1");
  HSK_REQUIRE((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g) & 15) == 0, HSK_ERR_INVALID,
              "buffers must be 16-byte aligned");
  if (n == 0) return HSK_OK;
  const hsk_adamw_consts c = hsk_make_adamw_consts(lr, beta1, beta2, eps, wd, step);
  const int64_t n4 = n >> 2;
  int64_t blocks = hsk_ceil_div(n4 > 0 ? n4 : 1, 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  k_adamw_dense<<<(unsigned)blocks, 256, 0, (hipStream_t)stream_>>>(p, g, m, v, (long long)n, c);
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}
