// hsk_item_sliced.h -- item-major gradient reduction + AdamW, sliced along D so that the gathered user rows stay in
// the L2 of the XCD that reads them.
//
// k_item_update (one wave per item, whole rows) gathers 2 KB user rows of the 4096 batch users = 8.4 MB, read by
// every XCD: ~half of the reads miss the 4 MB L2 and go to the Infinity Cache.  dI[i][d] has no coupling across d,
// so the pass can be cut into slices of 64 floats: a wave owns (item, slice), lane l owns element slice*64 + l.
// Workgroups are numbered so that blockIdx % 8 == slice % 8; blocks b and b+8 share an XCD (observed round-robin
// dispatch, a speed assumption only), so one XCD only ever touches 1/8 of every user row: 4096 x 256 B = 1 MB at
// D=512, resident in its L2.  Same bytes, L2 rate instead of Infinity-Cache rate.
#pragma once
#include "hsk_rows.h"
#include "hsk_step_kernels.h"
#include "hsk_fwd_part.h"

#ifndef HSK_ITEM_MLP
// user-row loads a wave keeps in flight.  4: with 8 the kernel sat on its 64-VGPR budget (8 waves per SIMD) with spills
// -- measured per step at the ml10m shape, three alternations on one box: 2 / 3 / 4: 188, 6: 190, 8: 194, 12: 238 us;
// 4 against 8 elsewhere: lfm2b-shaped training 509 against 560 us, the cfg5 share's item pass 6.26 against 6.38 ms
#define HSK_ITEM_MLP 4
#endif
#ifndef HSK_ITEM_NT
#define HSK_ITEM_NT 0      // 1: the item rows (p, m, v: read once, written once per step) move with non-temporal hints
#endif
template <int VS>
__device__ __forceinline__ hsk_vec<VS> hsk_ldg_stream(const float* p) {
#if HSK_ITEM_NT
  hsk_vec<VS> r;
#pragma unroll
  for (int i = 0; i < VS; ++i) r.v[i] = __builtin_nontemporal_load(p + i);
  return r;
#else
  return hsk_ldg<VS>(p);
#endif
}
template <int VS>
__device__ __forceinline__ void hsk_stg_stream(float* p, const hsk_vec<VS>& x) {
#if HSK_ITEM_NT
#pragma unroll
  for (int i = 0; i < VS; ++i) __builtin_nontemporal_store(x.v[i], p + i);
#else
  hsk_stg<VS>(p, x);
#endif
}

#ifdef HSK_DEBUG_XCC
__device__ unsigned hsk_dbg_xcc[64];   // [workgroup label (index % 8)][XCC id] of the item workgroups, debugging builds only
#endif
struct hsk_item_args {
  const float* Uw;        // user rows: the table (+ u32), an exchange buffer (+ slot index) or ucur (u32 == NULL)
  float* Iw; float* Ib; float* mI; float* vI; float* mIb; float* vIb;
  const int* u32;         // row of Uw for batch position e / K; NULL: the position itself
  const float* g_s; const int* perm; const int* offsets;
  int n_items, K, D, n_slices_pad, items_per_wave;
  hsk_adamw_consts c;
  float* gI_out; float* gIb_out;
  const int* touched; const int* n_touched; int* last_step_i; int step;   // LAZY only
  const int* pend;        // LAZY: per list entry, the step the row's MOMENTS stand at (k_item_catch_up wrote p only)
  const hsk_step_desc* desc; int rel;   // graph replay: step = desc->step0 + rel + 1, c from ctab
  const float2* ctab; int ctab_len;
  int n_part;             // PART kernels (item-partitioned forward): d loss/d s_0 of a positive = g_s[e] + .. + g_s[e + n_part-1]
};

// d loss / d score of entry e of batch position bpos.  PART: the forward's n_part units left their shares of the
// positive's weight in the first n_part columns of the row (column 0 is the positive's entry, the others are no entries)
template <bool PART>
__device__ __forceinline__ float hsk_entry_weight(const hsk_item_args& a, int e, int bpos) {
  float g = a.g_s[e];
  if (PART && e == bpos * a.K)
    for (int q = 1; q < a.n_part; ++q) g += a.g_s[e + q];
  return g;
}

// VS floats per lane: slice width = 64*VS floats; GEN: see hsk_adamw_update; LAZY: only the items in `touched`.
// `bid` = workgroup index inside the item pass (the launch may carry other workgroups in front, see k_item_user).
template <bool APPLY, int VS, bool GEN, bool LAZY, bool PART = false>
__device__ __forceinline__ void hsk_item_sliced_body(const hsk_item_args& a, int bid) {
  const float* __restrict__ Uw = a.Uw;
  float* __restrict__ Iw = a.Iw;
  float* __restrict__ Ib = a.Ib;
  float* __restrict__ mI = a.mI;
  float* __restrict__ vI = a.vI;
  float* __restrict__ mIb = a.mIb;
  float* __restrict__ vIb = a.vIb;
  const int* __restrict__ u32 = a.u32;
  const float* __restrict__ g_s = a.g_s;
  const int* __restrict__ perm = a.perm;
  const int* __restrict__ offsets = a.offsets;
  const int n_items = a.n_items, K = a.K, D = a.D, n_slices_pad = a.n_slices_pad, items_per_wave = a.items_per_wave;
  hsk_adamw_consts c = a.c;
  int step = a.step;
  hsk_resolve_step(a.desc, a.rel, a.ctab, a.ctab_len, step, c);
  float* __restrict__ gI_out = a.gI_out;
  float* __restrict__ gIb_out = a.gIb_out;
  const int* __restrict__ touched = a.touched;
  const int* __restrict__ n_touched = a.n_touched;
  int* __restrict__ last_step_i = a.last_step_i;
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  // n_slices_pad = number of slices when it divides 8 (each slice then owns 8/n XCDs), else a multiple of 8
  int slice, group;
  if (n_slices_pad < 8) {
    const int xcd = bid & 7, r = bid >> 3;
    slice = xcd % n_slices_pad;
    group = r * (8 / n_slices_pad) + xcd / n_slices_pad;
  } else {
    slice = bid % n_slices_pad;
    group = bid / n_slices_pad;
  }
  constexpr int SW = 64 * VS;
  const int d = slice * SW + lane * VS;
  const bool live = d < D;                       // the last slice may be partial (D % VS == 0); padded slices are empty
  if (slice * SW >= D) return;
  const int first = (group * 4 + wave) * items_per_wave;
  const int n_list = LAZY ? hsk_uniform_i(*n_touched) : n_items;
  for (int t = 0; t < items_per_wave; ++t) {
    const int idx = first + t;
    if (idx >= n_list) return;
    // lazy item AdamW: only the items that have entries are visited (their rows were brought up to step-1 by
    // k_item_catch_up); the others keep their zero-gradient steps for later
    const int i = LAZY ? hsk_uniform_i(touched[idx]) : idx;
    const int beg = hsk_uniform_i(offsets[i]);
    const int end = hsk_uniform_i(offsets[i + 1]);
    // AdamW operands early: their latency hides under the gather
    hsk_vec<VS> p = hsk_zero<VS>(), m = hsk_zero<VS>(), v = hsk_zero<VS>();
    if (APPLY && live) {
      p = hsk_ldg_stream<VS>(Iw + (long long)i * D + d);
      m = hsk_ldg_stream<VS>(mI + (long long)i * D + d);
      v = hsk_ldg_stream<VS>(vI + (long long)i * D + d);
    }
    hsk_vec<VS> acc = hsk_zero<VS>();
    float gb_lane = 0.f;
    for (int c0 = beg; c0 < end; c0 += 64) {
      const int nr = min(64, end - c0);
      int myu = 0;
      float myg = 0.f;
      if (lane < nr) {
        const int e = perm[c0 + lane];
        const int bpos = e / K;
        myg = PART ? hsk_entry_weight<true>(a, e, bpos) : g_s[e];
        myu = u32 ? u32[bpos] : bpos;   // NULL: user rows are laid out by batch position (ucur)
      }
      gb_lane += myg;
      for (int j = 0; j < nr; j += HSK_ITEM_MLP) {
        hsk_vec<VS> val[HSK_ITEM_MLP];
#pragma unroll
        for (int r = 0; r < HSK_ITEM_MLP; ++r)
          val[r] = (j + r < nr && live) ? hsk_ldg<VS>(Uw + (long long)hsk_readlane_i(myu, min(j + r, 63)) * D + d)
                                        : hsk_zero<VS>();
#pragma unroll
        for (int r = 0; r < HSK_ITEM_MLP; ++r)
          if (j + r < nr) {
            const float g = hsk_readlane_f(myg, j + r);
#pragma unroll
            for (int q = 0; q < VS; ++q) acc.v[q] = fmaf(g, val[r].v[q], acc.v[q]);
          }
      }
    }
    if (APPLY) {
      // lazy rows: the moments still stand at step pend[idx] (the catch-up wrote the parameter only): the missed
      // zero-gradient steps' m <- m + w1 (-m), v <- beta2 v, the replay's own operations
      const int behind = (LAZY && !GEN && a.pend) ? (step - 1) - hsk_uniform_i(a.pend[idx]) : 0;
      for (int r = 0; r < behind; ++r) {
#pragma unroll
        for (int q = 0; q < VS; ++q) {
          m.v[q] = fmaf(c.w1, -m.v[q], m.v[q]);
          v.v[q] = v.v[q] * c.beta2;
        }
      }
      if (live) {
#pragma unroll
        for (int q = 0; q < VS; ++q) hsk_adamw_update<GEN>(p.v[q], m.v[q], v.v[q], acc.v[q], c);
        hsk_stg_stream<VS>(Iw + (long long)i * D + d, p);
        hsk_stg_stream<VS>(mI + (long long)i * D + d, m);
        hsk_stg_stream<VS>(vI + (long long)i * D + d, v);
      }
      if (slice == 0 && Ib) {
        const float gbias = hsk_wave_sum(gb_lane);
        if (lane == 0) {
          float pb = Ib[i], mb = mIb[i], vb = vIb[i];
          for (int r = 0; r < behind; ++r) {
            mb = fmaf(c.w1, -mb, mb);
            vb = vb * c.beta2;
          }
          hsk_adamw_update<GEN>(pb, mb, vb, gbias, c);
          Ib[i] = pb;
          mIb[i] = mb;
          vIb[i] = vb;
        }
      }
      if (LAZY && slice == 0 && lane == 0) last_step_i[i] = step;
    } else {
      if (live) hsk_stg<VS>(gI_out + (long long)i * D + d, acc);
      if (slice == 0 && gIb_out) {
        const float gbias = hsk_wave_sum(gb_lane);
        if (lane == 0) gIb_out[i] = gbias;
      }
    }
  }
}

// Whole rows, one wave per item -- what a SMALL batch wants: with a few thousand entries everything is cache-resident
// and the pass is a chain of latencies per wave (offsets -> perm -> g_s -> rows), so the fewer waves the better; the
// D-slices above multiply them by the number of slices for an L2 affinity that only matters when the chip is full.
// LAZY: only the items in `touched`.
template <int V, int NCH, bool FULL, bool GEN, bool LAZY>
__device__ __forceinline__ void hsk_item_row_body(const hsk_item_args& a, int bid) {
  const int lane = hsk_lane();
  const int idx = bid * 4 + hsk_uniform_i(threadIdx.x >> 6);
  const int n_list = LAZY ? hsk_uniform_i(*a.n_touched) : a.n_items;
  if (idx >= n_list) return;
  hsk_adamw_consts c = a.c;
  int step = a.step;
  hsk_resolve_step(a.desc, a.rel, a.ctab, a.ctab_len, step, c);
  const int i = LAZY ? hsk_uniform_i(a.touched[idx]) : idx;
  const int D = a.D, K = a.K;
  using Row = hsk_row<V, NCH>;
  const int beg = hsk_uniform_i(a.offsets[i]);
  const int end = hsk_uniform_i(a.offsets[i + 1]);
  float* prow = a.Iw + (long long)i * D;
  float* mrow = a.mI + (long long)i * D;
  float* vrow = a.vI + (long long)i * D;
  Row p, m, v, acc;
  hsk_row_load<V, NCH, FULL>(p, prow, lane, D);   // AdamW operands early: their latency hides under the gather
  hsk_row_load<V, NCH, FULL>(m, mrow, lane, D);
  hsk_row_load<V, NCH, FULL>(v, vrow, lane, D);
  float pb = 0.f, mb = 0.f, vb = 0.f;   // the bias triple too (it used to be loaded after the reduction: a latency at the tail)
  if (a.Ib && lane == 0) {
    pb = a.Ib[i];
    mb = a.mIb[i];
    vb = a.vIb[i];
  }
  hsk_row_zero(acc);
  float gb_lane = 0.f;
  for (int c0 = beg; c0 < end; c0 += 64) {
    const int nr = min(64, end - c0);
    int myu = 0;
    float myg = 0.f;
    if (lane < nr) {
      const int e = a.perm[c0 + lane];
      myg = a.g_s[e];
      myu = a.u32 ? a.u32[e / K] : e / K;
    }
    gb_lane += myg;
    for (int j = 0; j < nr; j += 4) {
      Row buf[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (j + r < nr) hsk_row_load<V, NCH, FULL>(buf[r], a.Uw + (long long)hsk_readlane_i(myu, min(j + r, 63)) * D, lane, D);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (j + r < nr) hsk_row_axpy(acc, hsk_readlane_f(myg, j + r), buf[r]);
    }
  }
  const int behind = (LAZY && !GEN && a.pend) ? (step - 1) - hsk_uniform_i(a.pend[idx]) : 0;   // see hsk_item_sliced_body
  for (int r = 0; r < behind; ++r) {
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
      for (int q = 0; q < V; ++q) {
        m.c[cc].v[q] = fmaf(c.w1, -m.c[cc].v[q], m.c[cc].v[q]);
        v.c[cc].v[q] = v.c[cc].v[q] * c.beta2;
      }
    mb = fmaf(c.w1, -mb, mb);
    vb = vb * c.beta2;
  }
#pragma unroll
  for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
    for (int q = 0; q < V; ++q) hsk_adamw_update<GEN>(p.c[cc].v[q], m.c[cc].v[q], v.c[cc].v[q], acc.c[cc].v[q], c);
  hsk_row_store<V, NCH, FULL>(p, prow, lane, D);
  hsk_row_store<V, NCH, FULL>(m, mrow, lane, D);
  hsk_row_store<V, NCH, FULL>(v, vrow, lane, D);
  if (a.Ib) {
    const float gbias = hsk_wave_sum(gb_lane);
    if (lane == 0) {
      hsk_adamw_update<GEN>(pb, mb, vb, gbias, c);
      a.Ib[i] = pb;
      a.mIb[i] = mb;
      a.vIb[i] = vb;
    }
  }
  if (LAZY && lane == 0) a.last_step_i[i] = step;
}

template <bool APPLY, int VS, bool GEN, bool LAZY = false, bool PART = false>
__global__ __launch_bounds__(256) void k_item_update_sliced(hsk_item_args a, const int* __restrict__ g_ovf = nullptr,
                                                            const int* __restrict__ g_poison = nullptr) {
  if (APPLY && hsk_guard_skip(g_ovf, g_poison)) return;   // sharded step: see hsk_guard_skip
  hsk_item_sliced_body<APPLY, VS, GEN, LAZY, PART>(a, (int)blockIdx.x);
}

// whole rows as a launch of their own: the sharded step's item pass over the TOUCHED rows of a shard (lazy item AdamW)
template <int V, int NCH, bool FULL, bool GEN, bool LAZY>
__global__ __launch_bounds__(256) void k_item_update_rows(hsk_item_args a, const int* __restrict__ g_ovf = nullptr,
                                                          const int* __restrict__ g_poison = nullptr) {
  if (hsk_guard_skip(g_ovf, g_poison)) return;   // sharded step: see hsk_guard_skip
  hsk_item_row_body<V, NCH, FULL, GEN, LAZY>(a, (int)blockIdx.x);
}

// The item pass and the owners' user-row update in ONE launch: the first n_user_blocks workgroups (a multiple of 8, so
// that the item workgroups keep their XCD affinity) are k_user_update_lazy's, the rest k_item_update_sliced's.  The two
// are independent once the item pass reads the batch's user rows from `ucur` instead of the table the owners rewrite:
// one launch boundary less, and the short user workgroups finish under the item pass.
// A third kind of workgroup is interleaved with the item pass's: n_ahead_oct OCTETS of hsk_user_ahead_body, bringing
// the NEXT batch's user rows up to date under the item pass (hsk_step_kernels.h).  One octet of them follows every
// `stride` octets of item workgroups (whole octets, so the item workgroups keep blockIdx % 8 == their own index % 8,
// i.e. their XCD affinity): the replay is VALU work that should run beside the memory-bound item waves, not in front
// of them.
// Two flavours, two kernels: the row body's registers would cost the D-sliced, latency-bound flavour its occupancy
// (measured: 90 -> 106 us at the ml10m shape), and the three argument blocks push the scalar registers past what 8
// waves per SIMD allow -- hence the explicit occupancy request on the large-batch kernel.
// LAZYI (huge catalogues, lazy item AdamW): the launch is AdamW traffic on the touched item rows with the VALUs idle,
// so n_ahead_blocks workgroups of hsk_user_ahead_body ride in front of the item workgroups here too.
// PART: the gradient rows are sums of partial rows (item-partitioned forward, hsk_fwd_part.h); a separate instantiation,
// so that the default kernel's register allocation (scalar spills in particular) stays what it was.
template <int V, int NCH, bool FULL, int VS, bool GEN, bool LAZYI, bool PART = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_item_user(hsk_item_args ia, hsk_user_lazy_args ua, int n_user_blocks, int dense_users, hsk_ahead_args aa,
                 int n_ahead_blocks, hsk_ride_item ri = hsk_ride_item{}) {
  int bid = (int)blockIdx.x;
  if (PART) {   // preparation phases of later batches riding in this launch (hsk_fwd_part.h: hsk_ride_item)
    if (bid < ri.n_total) {
      __shared__ int ride_lds[HSK_SORT_SCATTER_LDS];
      if (bid < ri.C.n_blocks) {
        hsk_sort_scatter_body(ri.C.it32, ri.C.n_entries, ri.C.plan, ri.C.hist, ri.C.btot, ri.C.perm1, ri.C.bstart, bid, ride_lds);
      } else if (bid - ri.C.n_blocks < ri.H.n_blocks) {
        hsk_sort_hist_body(ri.H.it32, ri.H.n_entries, ri.H.plan, ri.H.hist, bid - ri.C.n_blocks, ride_lds);
      }
      return;
    }
    bid -= ri.n_total;
  }
  if (bid < n_user_blocks) {
    if (dense_users)
      hsk_user_update_dense_body<V, NCH, FULL, GEN, PART>(ua, bid);
    else
      hsk_user_update_lazy_body<V, NCH, FULL, GEN, PART>(ua, bid);
    return;
  }
  if (LAZYI && bid < n_user_blocks + n_ahead_blocks) {
    hsk_user_ahead_body<V, NCH, FULL, GEN>(aa, bid - n_user_blocks);
    return;
  }
#ifdef HSK_DEBUG_XCC
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    atomicAdd(&hsk_dbg_xcc[((bid - n_user_blocks) & 7) * 8 + (xcc & 7)], 1u);
  }
#endif
  hsk_item_sliced_body<true, VS, GEN, LAZYI, PART>(ia, bid - n_user_blocks - (LAZYI ? n_ahead_blocks : 0));
}

// small batches: whole-row item workgroups, interleaved with the ahead workgroups
template <int V, int NCH, bool FULL, bool GEN, bool LAZYI>
__global__ __launch_bounds__(256) void k_item_user_small(hsk_item_args ia, hsk_user_lazy_args ua, int n_user_blocks,
                                                         int dense_users, hsk_ahead_args aa, int n_ahead_oct,
                                                         int stride) {
  const int bid = (int)blockIdx.x;
  if (bid < n_user_blocks) {
    if (dense_users)
      hsk_user_update_dense_body<V, NCH, FULL, GEN>(ua, bid);
    else
      hsk_user_update_lazy_body<V, NCH, FULL, GEN>(ua, bid);
    return;
  }
  const int q = (bid - n_user_blocks) >> 3, r = bid & 7;   // n_user_blocks is a multiple of 8
  int item_oct = q;
  if (n_ahead_oct > 0) {
    const int period = stride + 1, full = n_ahead_oct * period;
    if (q < full) {
      const int k = q / period, j = q - k * period;
      if (j == stride) {
        hsk_user_ahead_body<V, NCH, FULL, GEN>(aa, k * 8 + r);
        return;
      }
      item_oct = k * stride + j;
    } else {
      item_oct = n_ahead_oct * stride + (q - full);
    }
  }
  hsk_item_row_body<V, NCH, FULL, GEN, LAZYI>(ia, item_oct * 8 + r);
}
