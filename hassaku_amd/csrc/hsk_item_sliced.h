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

template <bool APPLY, int VS, bool GEN, bool LAZY = false>   // VS floats per lane: slice width = 64*VS floats; GEN: see
                                                            // hsk_adamw_update; LAZY: only the items in `touched`
__global__ __launch_bounds__(256) void k_item_update_sliced(const float* __restrict__ Uw, float* __restrict__ Iw,
                                                            float* __restrict__ Ib, float* __restrict__ mI,
                                                            float* __restrict__ vI, float* __restrict__ mIb,
                                                            float* __restrict__ vIb, const int* __restrict__ u32,
                                                            const float* __restrict__ g_s, const int* __restrict__ perm,
                                                            const int* __restrict__ offsets, int n_items, int K, int D,
                                                            int n_slices_pad, int items_per_wave, hsk_adamw_consts c,
                                                            float* __restrict__ gI_out, float* __restrict__ gIb_out,
                                                            const int* __restrict__ touched = nullptr,
                                                            const int* __restrict__ n_touched = nullptr,
                                                            int* __restrict__ last_step_i = nullptr, int step = 0) {
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  // n_slices_pad = number of slices when it divides 8 (each slice then owns 8/n XCDs), else a multiple of 8
  int slice, group;
  if (n_slices_pad < 8) {
    const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
    slice = xcd % n_slices_pad;
    group = r * (8 / n_slices_pad) + xcd / n_slices_pad;
  } else {
    slice = blockIdx.x % n_slices_pad;
    group = blockIdx.x / n_slices_pad;
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
      p = hsk_ldg<VS>(Iw + (long long)i * D + d);
      m = hsk_ldg<VS>(mI + (long long)i * D + d);
      v = hsk_ldg<VS>(vI + (long long)i * D + d);
    }
    hsk_vec<VS> acc = hsk_zero<VS>();
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
      for (int j = 0; j < nr; j += 8) {
        hsk_vec<VS> val[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
          val[r] = (j + r < nr && live) ? hsk_ldg<VS>(Uw + (long long)hsk_readlane_i(myu, min(j + r, 63)) * D + d)
                                        : hsk_zero<VS>();
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if (j + r < nr) {
            const float g = hsk_readlane_f(myg, j + r);
#pragma unroll
            for (int q = 0; q < VS; ++q) acc.v[q] = fmaf(g, val[r].v[q], acc.v[q]);
          }
      }
    }
    if (APPLY) {
      if (live) {
#pragma unroll
        for (int q = 0; q < VS; ++q) hsk_adamw_update<GEN>(p.v[q], m.v[q], v.v[q], acc.v[q], c);
        hsk_stg<VS>(Iw + (long long)i * D + d, p);
        hsk_stg<VS>(mI + (long long)i * D + d, m);
        hsk_stg<VS>(vI + (long long)i * D + d, v);
      }
      if (slice == 0 && Ib) {
        const float gbias = hsk_wave_sum(gb_lane);
        if (lane == 0) {
          float pb = Ib[i], mb = mIb[i], vb = vIb[i];
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
