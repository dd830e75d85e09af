// hsk_step_kernels.h -- the three heavy kernels of the fused BPR-MF step and its tail.
#pragma once
#include "hsk_rows.h"

#ifndef HSK_REPLAY_NT
#define HSK_REPLAY_NT 0   // 1 / 2: moments / whole rows of replayed rows leave with non-temporal stores (hsk_stg_nt).
                          // Measured at the hbm shape (forward behind the lazy item catch-up): 216 -> 210 us forward,
                          // the catch-up itself slower: off.
#endif

__device__ __forceinline__ hsk_adamw_consts hsk_consts_at(const hsk_adamw_consts& base, const float2* __restrict__ tab,
                                                          int tab_len, int t) {
  hsk_adamw_consts c = base;
  const float2 e = tab[min(t, tab_len)];   // (step_size, bc2_sqrt or its reciprocal, see hsk_fused.hip)
  c.step_size = e.x;
#if HSK_ADAM_IEEE
  c.bc2_sqrt = e.y;
#else
  c.rbc2_sqrt = e.y;
#endif
  return c;
}


// The per-step scalars of up to 64 consecutive steps t0 .. t0+63 in ONE vector load (lane j: step t0 + j), handed out by
// readlane: a table load per replayed step puts a memory latency into every iteration of the serial replay chain.
__device__ __forceinline__ float2 hsk_consts_block(const float2* __restrict__ tab, int tab_len, int t0) {
  return tab[min(t0 + hsk_lane(), tab_len)];
}
__device__ __forceinline__ hsk_adamw_consts hsk_consts_lane(const hsk_adamw_consts& base, float2 block, int j) {
  hsk_adamw_consts c = base;
  c.step_size = hsk_readlane_f(block.x, j);
#if HSK_ADAM_IEEE
  c.bc2_sqrt = hsk_readlane_f(block.y, j);
#else
  c.rbc2_sqrt = hsk_readlane_f(block.y, j);
#endif
  return c;
}

// ---------------------------------------------------------------------------------------------
// Sharded step (hsk_shard.inc): a batch that exceeded a capacity must never update a table.  k_shard_place raises *ovf
// (one word per buffer set) for such a batch; the kernel that OPENS its step (k_user_catch_up, launched by hsk_shard_pack)
// then raises the sticky *poison, and every table-writing kernel of this and of all later steps returns at once: the
// tables stay what they were before the first overflowing step until the host reads HSK_STATUS_SHARD_OVERFLOW and raises.
// One scalar load (two) per workgroup.  NULL pointers (the single-GPU step): no guard.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool hsk_guard_skip(const int* __restrict__ ovf, const int* __restrict__ poison) {
  return (ovf && __builtin_nontemporal_load(ovf) != 0) || (poison && __builtin_nontemporal_load(poison) != 0);
}

// ---------------------------------------------------------------------------------------------
// Lazy user rows inside the forward.  A user row outside the previous batches carries pending zero-gradient AdamW
// steps (see "Lazy, exact user-table AdamW" below).  Instead of a separate catch-up launch that rewrites the row
// before the forward reads it, the forward replays the missed steps IN REGISTERS on its own copy (p, m, v of the row:
// two more row loads for the rows that are behind), scores with the current row and leaves it in `ucur[b]` for the
// item pass.  The table itself is not touched here (duplicate entries of a user replay the same stale row and must
// not see a half-rewritten one): the owner entry also leaves the replayed moments in `mcur[b]` / `vcur[b]`, and the
// owner's update kernel picks the current (p, m, v) up from there instead of replaying again.  Non-owner entries of a
// user (duplicates in the batch) register with the owner.
// ---------------------------------------------------------------------------------------------
struct hsk_lazy_user_args {
  const float* mU;        // NULL: rows are current (dense user updates, or rows handed over by an exchange)
  const float* vU;
  const int* last_step;
  const int* owner;
  int* dupcnt;
  int* duplist;
  float* ucur;            // [B, D]: the batch's user rows as the scores saw them
  float* mcur;            // [B, D]: replayed moments of the rows that were behind (owner entries only)
  float* vcur;
  int step;               // the step being taken
  int gen;                // 1: generic optimiser arithmetic (adam / adagrad), 0: AdamW
  hsk_adamw_consts c;
  const float2* tab;
  int tab_len;
  const hsk_step_desc* desc;   // graph replay: step = desc->step0 + rel + 1
  int rel;
};

#define HSK_DUP_MAX 8   // duplicate entries of one user that are listed for the owner (more: batch scan)

// `done` / `own`: last_step[u] and owner[u], loaded by the caller (early, together with its other loads)
template <int V, int NCH, bool FULL>
__device__ __forceinline__ void hsk_user_row_current(hsk_row<V, NCH>& p, int u, int b, int B, int D, int lane,
                                                     const hsk_lazy_user_args& lz, bool publish, int done, int own) {
  const int step = lz.desc ? lz.desc->step0 + lz.rel + 1 : lz.step;
  if (done < step - 1) {
    hsk_row<V, NCH> m, v;
    hsk_row_load<V, NCH, FULL>(m, lz.mU + (long long)u * D, lane, D);
    hsk_row_load<V, NCH, FULL>(v, lz.vU + (long long)u * D, lane, D);
    for (int t0 = done + 1; t0 <= step - 1; t0 += 64) {
      const float2 blk = hsk_consts_block(lz.tab, lz.tab_len, t0);
      const int nt = min(64, step - t0);
      for (int j = 0; j < nt; ++j) {
        const hsk_adamw_consts ct = hsk_consts_lane(lz.c, blk, j);
        if (lz.gen) {
#pragma unroll
          for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
            for (int q = 0; q < V; ++q) hsk_adamw_replay<true>(p.c[cc].v[q], m.c[cc].v[q], v.c[cc].v[q], ct);
        } else {
#pragma unroll
          for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
            for (int q = 0; q < V; ++q) hsk_adamw_replay<false>(p.c[cc].v[q], m.c[cc].v[q], v.c[cc].v[q], ct);
        }
      }
    }
    if (publish && own == b) {
      hsk_row_store<V, NCH, FULL>(m, lz.mcur + (long long)b * D, lane, D);
      hsk_row_store<V, NCH, FULL>(v, lz.vcur + (long long)b * D, lane, D);
    }
  }
  if (!publish) return;
  hsk_row_store<V, NCH, FULL>(p, lz.ucur + (long long)b * D, lane, D);
  if (lane == 0) {
    if (own != b && own >= 0 && own < B) {   // a further entry of a user somebody else owns: tell the owner
      const int slot = atomicAdd(&lz.dupcnt[own], 1);
      if (slot < HSK_DUP_MAX) lz.duplist[own * HSK_DUP_MAX + slot] = b;
    }
  }
}

// =============================================================================================
// K1: per positive b -- gather u row + (1+N) item rows, scores, BPR loss terms, d loss/d score,
//     user-row gradient (accumulated in registers).  One wave per positive.
//     reads:  4*D*(2+N) + 4*(1+N) + 4*(2+N) bytes per positive  (the "gather+BPR step" of SURVEY 8d)
//     writes: g_s [B,1+N], dUb [B,D], loss_b [B]
// =============================================================================================
// LOSS selects the recommendation loss evaluated on the (1+N) scores of a positive (train/rec_losses.py):
//   HSK_LOSS_BPR  mean_{b,n} softplus(-(s_0 - s_n))                                   (:56-88)
//   HSK_LOSS_BCE  mean_{b,k} BCEWithLogits(s_k, [k==0])                               (:27-53)
//   HSK_LOSS_SSM  mean_b  -s_0 + logsumexp(s_0, s_1 + c, ..., s_N + c), c = log(I/N)  (:91-139, uniform sampling)
// All three need every row once: BPR and BCE weigh a row as soon as its score (and s_0) is known; the sampled
// softmax keeps a running max / normaliser and rescales the accumulated user-row gradient when the max moves
// (online softmax), then rewrites its stored scores into gradients once the normaliser is final.
// (HSK_LOSS_BPR / _BCE / _SSM are declared in hassaku_hip.h)

template <int V, int NCH, bool FULL, int R, int LOSS>
__global__ __launch_bounds__(256) void k_fwd_ugrad(const float* __restrict__ Uw, const float* __restrict__ Iw,
                                                   const float* __restrict__ Ib, const int* __restrict__ u32,
                                                   const int* __restrict__ it32, int B, int K, int D,
                                                   float inv_norm, float ssm_c, float* __restrict__ g_s,
                                                   float* __restrict__ dUb, double* __restrict__ loss_b,
                                                   const int* __restrict__ dU_index = nullptr,
                                                   hsk_lazy_user_args lz = hsk_lazy_user_args{}) {
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;

  using Row = hsk_row<V, NCH>;
  const int* __restrict__ irow = it32 + (long long)b * K;
  // Short rows (D <= 256) mean small tables and small batches: the kernel is a chain of memory latencies, so the loads are
  // issued in two rounds instead of four -- (1) user id, positive and the first 64 item ids, (2) user row, positive's row,
  // biases and the first R item rows.  (Long rows: the early item rows would cost the large-batch shapes a wave per SIMD.)
  constexpr bool EARLY = NCH * V <= 4;
  const int u_v = u32[b];
  const int i0_v = irow[0];
  int id_first = (EARLY && lane < min(64, K - 1)) ? irow[1 + lane] : -1;
  const int u = hsk_uniform_i(u_v);
  const int i0 = hsk_uniform_i(i0_v);

  Row ur, r0, acc;
  hsk_row_load<V, NCH, FULL>(ur, Uw + (long long)u * D, lane, D);
  hsk_row_load<V, NCH, FULL>(r0, Iw + (long long)i0 * D, lane, D);
  const float bias0 = Ib ? Ib[i0] : 0.f;
  float bias_first = 0.f;
  Row bufA[R], bufB[R];
  if (EARLY) {
    if (id_first < 0) id_first = i0;
    bias_first = Ib ? Ib[id_first] : 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r < K - 1) hsk_row_load<V, NCH, FULL>(bufA[r], Iw + (long long)hsk_readlane_i(id_first, r) * D, lane, D);
  }
  if (lz.mU)
    hsk_user_row_current<V, NCH, FULL>(ur, u, b, B, D, lane, lz, true, hsk_uniform_i(lz.last_step[u]),
                                       hsk_uniform_i(lz.owner[u]));
  else if (lz.ucur)   // dense user updates: the rows are current; the item pass still reads them by batch position
    hsk_row_store<V, NCH, FULL>(ur, lz.ucur + (long long)b * D, lane, D);
  hsk_row_zero(acc);
  const float s0 = hsk_wave_sum(hsk_row_dot_partial(ur, r0)) + bias0;

  float gsum = 0.f;    // BPR: sum over negatives of sigma(-x)/(B*N)   (wave-uniform)
  double lsum = 0.0;   // per-lane partial of the loss terms
  float smax = s0, ssum = 1.f;   // SSM: running max and sum of exp(z - smax), z_0 = s_0
  if (LOSS == HSK_LOSS_SSM) acc = r0;  // exp(z_0 - smax) = 1

  for (int kc = 1; kc < K; kc += 64) {
    const int nr = min(64, K - kc);
    const bool first = EARLY && kc == 1;   // ids, biases and the first rows of this chunk are already on their way
    const int myidx = first ? id_first : (lane < nr) ? irow[kc + lane] : i0;
    const float mybias = first ? bias_first : Ib ? Ib[myidx] : 0.f;
    float gv = 0.f, xv = 0.f;

    // prologue
    if (!first) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (r < nr) hsk_row_load<V, NCH, FULL>(bufA[r], Iw + (long long)hsk_readlane_i(myidx, r) * D, lane, D);
    }

    auto process = [&](Row(&buf)[R], int j) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (j + r < nr) {
          const float s = hsk_wave_sum(hsk_row_dot_partial(ur, buf[r])) + hsk_readlane_f(mybias, j + r);
          if (LOSS == HSK_LOSS_BPR) {
            const float x = s0 - s;
            const float g = inv_norm / (1.f + expf(x));  // sigma(-x)/(B*N) = d loss / d s_neg
            hsk_row_axpy(acc, g, buf[r]);
            gsum += g;
            gv = (lane == j + r) ? g : gv;
            xv = (lane == j + r) ? x : xv;
          } else if (LOSS == HSK_LOSS_BCE) {
            const float g = inv_norm / (1.f + expf(-s));  // sigma(s)/(B*K), label 0
            hsk_row_axpy(acc, g, buf[r]);
            gv = (lane == j + r) ? g : gv;
            xv = (lane == j + r) ? s : xv;
          } else {
            const float z = s + ssm_c;
            if (z > smax) {  // wave-uniform branch: rescale what has been accumulated so far
              const float f = expf(smax - z);
              ssum *= f;
#pragma unroll
              for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
                for (int q = 0; q < V; ++q) acc.c[cc].v[q] *= f;
              smax = z;
            }
            const float e = expf(z - smax);
            ssum += e;
            hsk_row_axpy(acc, e, buf[r]);
            gv = (lane == j + r) ? z : gv;  // finalised below
          }
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
      if (LOSS == HSK_LOSS_BPR) lsum += (double)hsk_softplus(-xv);
      if (LOSS == HSK_LOSS_BCE) lsum += (double)hsk_softplus(xv);
    }
  }
  float g0;
  if (LOSS == HSK_LOSS_BPR) {
    g0 = -gsum;                       // d loss / d s_pos = -sum_n sigma(-x_n)/(B*N)
    hsk_row_axpy(acc, g0, r0);
  } else if (LOSS == HSK_LOSS_BCE) {
    g0 = -inv_norm / (1.f + expf(s0));   // (sigma(s_0) - 1)/(B*K)
    hsk_row_axpy(acc, g0, r0);
    if (lane == 0) lsum += (double)hsk_softplus(-s0);
  } else {
    // softmax_k = exp(z_k - smax)/ssum;  d loss/d s_k = (softmax_k - [k==0]) / B
    const float rinv = inv_norm / ssum;
    g0 = expf(s0 - smax) * rinv - inv_norm;
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
      for (int q = 0; q < V; ++q) acc.c[cc].v[q] = fmaf(acc.c[cc].v[q], rinv, -inv_norm * r0.c[cc].v[q]);
    for (int k = 1 + lane; k < K; k += 64) {   // stored z_k -> gradients (same lanes wrote them)
      const float z = g_s[(long long)b * K + k];
      g_s[(long long)b * K + k] = expf(z - smax) * rinv;
    }
    if (lane == 0) lsum += (double)(-s0 + smax + logf(ssum));
  }
  if (lane == 0) g_s[(long long)b * K] = g0;
  // row-sharded user tables: the gradient row goes straight into its slot of the all-to-all send buffer
  hsk_row_store<V, NCH, FULL>(acc, dUb + (long long)(dU_index ? hsk_uniform_i(dU_index[b]) : b) * D, lane, D);
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
                                                     const int* __restrict__ offsets,
                                                     int n_items, int K, int D, hsk_adamw_consts c,
                                                     float* __restrict__ gI_out, float* __restrict__ gIb_out,
                                                     const int* __restrict__ g_ovf = nullptr,
                                                     const int* __restrict__ g_poison = nullptr) {
  if (APPLY && hsk_guard_skip(g_ovf, g_poison)) return;
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
      myu = u32 ? u32[e / K] : e / K;   // NULL: user rows are laid out by batch position (ucur)
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
    }
  } else {
    hsk_row_store<V, NCH, FULL>(acc, gI_out + (long long)i * D, lane, D);
    if (lane == 0) {
      if (gIb_out) gIb_out[i] = gbias;
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
                                              int D, int lane, int* __restrict__ dupcnt = nullptr,
                                              const int* __restrict__ duplist = nullptr) {
  hsk_row_load<V, NCH, FULL>(acc, dUb + (long long)b0 * D, lane, D);
  if (c > 1 && dupcnt) {
    // the non-owner entries registered themselves with the owner (k_user_catch_up), in arrival order: sort the few
    // of them so that the gradient rows are summed in ascending b, whatever the arrival order was
    const int nd = hsk_uniform_i(dupcnt[b0]);
    if (lane == 0) dupcnt[b0] = 0;
    if (nd == c - 1 && nd <= HSK_DUP_MAX) {
      const int mine = (lane < nd) ? duplist[b0 * HSK_DUP_MAX + lane] : 0x7fffffff;
      int rank = 0;
#pragma unroll
      for (int k = 0; k < HSK_DUP_MAX; ++k) rank += (k < nd && hsk_readlane_i(mine, k) < mine) ? 1 : 0;
      for (int r = 0; r < nd; ++r) {
        const unsigned long long m = __ballot(lane < nd && rank == r);
        const int bb = hsk_readlane_i(mine, __builtin_ctzll(m));
        hsk_row<V, NCH> t;
        hsk_row_load<V, NCH, FULL>(t, dUb + (long long)bb * D, lane, D);
        hsk_row_add(acc, t);
      }
      return;
    }
  }
  if (c > 1) {
    // duplicates of this user further down the batch, in ascending b; the id loads of 8 chunks (512 entries) are
    // issued together so the scan costs a handful of memory latencies, not one per chunk
    int found = 1;
    for (int g0 = (b0 / 64) * 64; g0 < B && found < c; g0 += 512) {
      int ids[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int bb = g0 + q * 64 + lane;
        ids[q] = (bb < B && bb > b0) ? u32[bb] : -1;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        unsigned long long mask = __ballot(ids[q] == row);
        while (mask) {
          const int j = __builtin_ctzll(mask);
          mask &= mask - 1;
          hsk_row<V, NCH> t;
          hsk_row_load<V, NCH, FULL>(t, dUb + (long long)(g0 + q * 64 + j) * D, lane, D);
          hsk_row_add(acc, t);
          ++found;
        }
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

// ---------------------------------------------------------------------------------------------
// Lazy, exact user-table AdamW.  torch.optim.AdamW updates EVERY row each step; a row outside the batch
// sees g = 0: p *= decay; m *= b1 (as lerp); v *= b2; p -= ss_t * m / (sqrt(v)/bc2s_t + eps).  That
// recurrence needs nothing but the row itself and the per-step scalars (ss_t, bc2s_t), so it can be
// replayed later: last_step[row] = number of steps already applied; when the row is next touched (or at a
// flush) the missed steps are replayed in registers with the same fp32 operations the dense sweep would
// have executed -- results are bit-identical to the dense sweep, HBM traffic is only the touched rows.
// tab[t] = (ss_t, bc2s_t) for t in [1, tab_len]; for t > tab_len the entry tab_len applies (both saturated).
// ---------------------------------------------------------------------------------------------
// Zero-gradient replay of one user row by a whole 256-thread workgroup (thread t owns VV consecutive elements per
// pass): the replay is a serial chain of `to - from` dependent updates per element, so a row is spread over as
// many lanes as it has elements instead of being held by one wave.
// P_ONLY: only the parameter leaves (the moments stay as of step `from`: whoever rewrites the row next replays them --
// two multiplications per element and step, no table -- see k_item_catch_up)
template <int VV, bool GEN, bool P_ONLY = false>
__device__ __forceinline__ void hsk_row_replay_wg(float* __restrict__ prow, float* __restrict__ mrow,
                                                  float* __restrict__ vrow, int D, int from, int to,
                                                  const hsk_adamw_consts& base, const float2* __restrict__ tab,
                                                  int tab_len) {
  for (int db = 0; db < D; db += 256 * VV) {   // uniform trip count: every lane takes part in the scalar hand-out below
    const int d0 = db + threadIdx.x * VV;
    const bool live = d0 < D;
    hsk_vec<VV> p = hsk_zero<VV>(), m = hsk_zero<VV>(), v = hsk_zero<VV>();
    if (live) {
      p = hsk_ldg<VV>(prow + d0);
      m = hsk_ldg<VV>(mrow + d0);
      v = hsk_ldg<VV>(vrow + d0);
    }
    for (int t0 = from + 1; t0 <= to; t0 += 64) {
      const float2 blk = hsk_consts_block(tab, tab_len, t0);
      const int nt = min(64, to - t0 + 1);
      for (int j = 0; j < nt; ++j) {
        const hsk_adamw_consts c = hsk_consts_lane(base, blk, j);
#pragma unroll
        for (int q = 0; q < VV; ++q) hsk_adamw_replay<GEN>(p.v[q], m.v[q], v.v[q], c);
      }
    }
    if (live) {
#if HSK_REPLAY_NT > 1
      hsk_stg_nt<VV>(prow + d0, p);
#else
      hsk_stg<VV>(prow + d0, p);
#endif
      if (!P_ONLY) {
#if HSK_REPLAY_NT
        hsk_stg_nt<VV>(mrow + d0, m);
        hsk_stg_nt<VV>(vrow + d0, v);
#else
        hsk_stg<VV>(mrow + d0, m);
        hsk_stg<VV>(vrow + d0, v);
#endif
      }
    }
  }
}

// Lazy mode, before the forward: one workgroup per batch entry; the owner (lowest b) of each distinct user brings
// that user's row up to step-1, so the gather/score kernels read current parameters.
template <int VV, bool GEN>
__global__ __launch_bounds__(256) void k_user_catch_up(float* __restrict__ Uw, float* __restrict__ mU,
                                                       float* __restrict__ vU, float* __restrict__ Ub,
                                                       float* __restrict__ mUb, float* __restrict__ vUb,
                                                       const int* __restrict__ u32, const int* __restrict__ owner,
                                                       int* __restrict__ last_step, int B, int D, int step,
                                                       hsk_adamw_consts c, const float2* __restrict__ tab,
                                                       int tab_len, int* __restrict__ dupcnt = nullptr,
                                                       int* __restrict__ duplist = nullptr,
                                                       const int* __restrict__ g_ovf = nullptr,
                                                       int* __restrict__ g_poison = nullptr, int g_open = 0) {
  const int b = blockIdx.x;
  if (g_poison) {   // sharded step (hsk_guard_skip); g_open: this launch opens the step of the batch *g_ovf speaks of
    const bool over = g_ovf && __builtin_nontemporal_load(g_ovf) != 0;
    if (over && g_open && b == 0 && threadIdx.x == 0) *g_poison = 1;
    if (over || __builtin_nontemporal_load(g_poison) != 0) return;
  }
  const int row = u32[b];
  if (row < 0) return;  // empty exchange slot (row-sharded mode)
  // row loads issued together with the owner / last_step lookups; thrown away by non-owners and current rows
  float* prow = Uw + (long long)row * D;
  float* mrow = mU + (long long)row * D;
  float* vrow = vU + (long long)row * D;
  const int d0 = threadIdx.x * VV;
  hsk_vec<VV> p0 = hsk_zero<VV>(), m0 = hsk_zero<VV>(), v0 = hsk_zero<VV>();
  if (d0 < D) {
    p0 = hsk_ldg<VV>(prow + d0);
    m0 = hsk_ldg<VV>(mrow + d0);
    v0 = hsk_ldg<VV>(vrow + d0);
  }
  const int own = owner[row];
  const int done = last_step[row];
  if (own != b) {
    // a further entry of a user somebody else owns: tell the owner, who sums the gradient rows after the forward
    if (dupcnt && threadIdx.x == 0 && own >= 0 && own < B) {
      const int slot = atomicAdd(&dupcnt[own], 1);
      if (slot < HSK_DUP_MAX) duplist[own * HSK_DUP_MAX + slot] = b;
    }
    return;
  }
  if (done >= step - 1) return;
  // first pass of the replay on the preloaded elements (every lane walks the steps: the scalars come by readlane)
  for (int t0 = done + 1; t0 <= step - 1; t0 += 64) {
    const float2 blk = hsk_consts_block(tab, tab_len, t0);
    const int nt = min(64, step - t0);
    for (int j = 0; j < nt; ++j) {
      const hsk_adamw_consts ct = hsk_consts_lane(c, blk, j);
#pragma unroll
      for (int q = 0; q < VV; ++q) hsk_adamw_replay<GEN>(p0.v[q], m0.v[q], v0.v[q], ct);
    }
  }
  if (d0 < D) {
    hsk_stg<VV>(prow + d0, p0);
    hsk_stg<VV>(mrow + d0, m0);
    hsk_stg<VV>(vrow + d0, v0);
  }
  if (D > 256 * VV)   // rows longer than one pass of the workgroup
    hsk_row_replay_wg<VV, GEN>(prow + 256 * VV, mrow + 256 * VV, vrow + 256 * VV, D - 256 * VV, done, step - 1, c, tab,
                               tab_len);
  __syncthreads();  // every thread has read last_step[row]
  if (threadIdx.x == 0) {
    if (Ub) {
      float pb = Ub[row], mb = mUb[row], vb = vUb[row];
      for (int t = done + 1; t <= step - 1; ++t) hsk_adamw_update<GEN>(pb, mb, vb, 0.f, hsk_consts_at(c, tab, tab_len, t));
      Ub[row] = pb;
      mUb[row] = mb;
      vUb[row] = vb;
    }
    last_step[row] = step - 1;
  }
}

// Lazy ITEM AdamW (catalogues far larger than a batch touches): the items of this batch -- the compact list the item
// sort leaves in `touched` -- are brought up to step-1 before the forward reads them; every other item row keeps its
// zero-gradient steps for later, exactly like the user rows above.  One workgroup per listed item.
// The forward needs the PARAMETER row only, and the item pass of this same step rewrites (p, m, v) of exactly these rows
// anyway: under AdamW (!GEN; with L2 decay the moments depend on p's path) only p is written here -- 2 of the 6 KB per
// row at D = 512 -- and pend[j] tells the item pass from which step the moments of list entry j are to be brought
// forward (m <- m + w1 (-m), v <- beta2 v per missed step: the replay's own operations, so the bits are the dense ones).
template <int VV, bool GEN>
__global__ __launch_bounds__(256) void k_item_catch_up(float* __restrict__ Iw, float* __restrict__ mI,
                                                       float* __restrict__ vI, float* __restrict__ Ib,
                                                       float* __restrict__ mIb, float* __restrict__ vIb,
                                                       const int* __restrict__ touched,
                                                       const int* __restrict__ n_touched, int* __restrict__ last_step_i,
                                                       int D, int step, hsk_adamw_consts c,
                                                       const float2* __restrict__ tab, int tab_len,
                                                       const hsk_step_desc* __restrict__ desc = nullptr, int rel = 0,
                                                       int* __restrict__ pend = nullptr,
                                                       const int* __restrict__ g_ovf = nullptr,
                                                       const int* __restrict__ g_poison = nullptr) {
  if (hsk_guard_skip(g_ovf, g_poison)) return;
  if (desc) step = desc->step0 + rel + 1;   // graph replay (c's per-step fields are rebuilt per replayed step below)
  if ((int)blockIdx.x >= *n_touched) return;
  const int row = touched[blockIdx.x];
  const int done = last_step_i[row];
  constexpr bool P_ONLY = !GEN;
  const bool p_only = P_ONLY && pend != nullptr;
  if (pend && threadIdx.x == 0) pend[blockIdx.x] = p_only ? min(done, step - 1) : step - 1;
  if (done >= step - 1) return;
  if (p_only)
    hsk_row_replay_wg<VV, GEN, P_ONLY>(Iw + (long long)row * D, mI + (long long)row * D, vI + (long long)row * D, D, done,
                                       step - 1, c, tab, tab_len);
  else
    hsk_row_replay_wg<VV, GEN>(Iw + (long long)row * D, mI + (long long)row * D, vI + (long long)row * D, D, done,
                               step - 1, c, tab, tab_len);
  __syncthreads();  // every thread has read last_step_i[row]
  if (threadIdx.x == 0) {
    if (Ib) {
      float pb = Ib[row], mb = mIb[row], vb = vIb[row];
      for (int t = done + 1; t <= step - 1; ++t) hsk_adamw_update<GEN>(pb, mb, vb, 0.f, hsk_consts_at(c, tab, tab_len, t));
      Ib[row] = pb;
      if (!p_only) {
        mIb[row] = mb;
        vIb[row] = vb;
      }
    }
    last_step_i[row] = step - 1;   // (of p; with pend set, of the moments as recorded there)
  }
}

// Tail work of a step that rides on the last launch as one extra workgroup (saves a dependent launch): deterministic
// fp64 tree sum of the per-positive loss terms and the zero-gradient AdamW step of the global bias.
struct hsk_finish_args {
  const double* loss_b;   // NULL: nothing to do
  int n;
  double inv_norm;
  double* loss_out;
  float* gb;
  float* mgb;
  float* vgb;
};

__device__ __forceinline__ void hsk_finish_block(const hsk_finish_args& f, const hsk_adamw_consts& c) {
  __shared__ double red[256];
  const int t = threadIdx.x;
  double s = 0.0;
  for (int i = t; i < f.n; i += 256) s += f.loss_b[i];
  red[t] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (t < off) red[t] += red[t + off];
    __syncthreads();
  }
  if (t == 0) {
    const double loss = red[0] * f.inv_norm;
    if (f.loss_out) {
      f.loss_out[0] = loss;
      f.loss_out[1] += loss;
    }
    if (f.gb) {
      float p = f.gb[0], m = f.mgb[0], v = f.vgb[0];
      hsk_adamw_update(p, m, v, 0.f, c);
      f.gb[0] = p;
      f.mgb[0] = m;
      f.vgb[0] = v;
    }
  }
}

// Lazy mode, after the gradients: the owner brings its row up to step-1 (pending zero-gradient steps; `tab` == NULL:
// the row is current already) and applies step `step`.  One extra workgroup (bid == ceil(B/4)) runs hsk_finish_block
// when `fin.loss_b` is set.  Body + argument block so that the item pass can carry these workgroups in its own launch.
struct hsk_user_lazy_args {
  float* Uw; float* mU; float* vU; float* Ub; float* mUb; float* vUb;
  const float* dUb;       // [B, D] gradient rows by batch position
  const int* u32;         // [B] table row of each batch position (-1: empty exchange slot)
  int* owner; int* cnt; int* last_step;
  int B, D, step;
  hsk_adamw_consts c;
  hsk_finish_args fin;
  int* dupcnt; const int* duplist;
  const float2* tab; int tab_len;   // tab == NULL: the table rows are current
  const float* ucur; const float* mcur; const float* vcur;   // else: (p, m, v) of the rows that are behind, by batch
                                                             // position, as the forward replayed them
  int n_users;            // dense sweep (hsk_user_update_dense_body): rows of the table
  const hsk_step_desc* desc; int rel;   // graph replay: step = desc->step0 + rel + 1, c from ctab
  const float2* ctab; int ctab_len;
  int n_part;   // item-partitioned forward: a gradient row is the sum of n_part partial rows, laid out [b][q][D]
                // (the n_part rows of a batch position side by side)
};

// gradient chunk of batch position b: the partial rows added in partition order
template <int V>
__device__ __forceinline__ hsk_vec<V> hsk_grad_chunk(const hsk_user_lazy_args& a, int b, int off) {
  const float* src = a.dUb + (long long)b * a.n_part * a.D + off;
  hsk_vec<V> g = hsk_ldg<V>(src);
  for (int q0 = 1; q0 < a.n_part; q0 += 3) {   // three loads in flight, then their sum in partition order
    hsk_vec<V> t[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
      t[j] = (q0 + j < a.n_part) ? hsk_ldg<V>(src + (long long)(q0 + j) * a.D) : hsk_zero<V>();
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < V; ++i) g.v[i] += t[j].v[i];
  }
  return g;
}

// step index and optimiser scalars of this launch: the immediate ones, or (graph replay) derived on the device
__device__ __forceinline__ void hsk_resolve_step(const hsk_step_desc* desc, int rel, const float2* ctab, int ctab_len,
                                                 int& step, hsk_adamw_consts& c) {
  if (desc) {
    step = desc->step0 + rel + 1;
    c = hsk_consts_at(c, ctab, ctab_len, step);
  }
}

// AdamW on one user row, one 64-lane chunk at a time ((p, m, v, g) of a whole row in registers would cost the item
// workgroups that share the launch their occupancy).  Gradient = dUb[b] + the rows of the user's further entries in
// the batch (n entries in all), added in ascending b: usually those registered with the owner (dupcnt / duplist, in
// arrival order: ranked here), otherwise the batch is scanned.  n == 0: no gradient (dense sweep over an idle row).
template <int V, int NCH, bool FULL, bool GEN, bool PART = false>
__device__ __forceinline__ void hsk_user_row_chunks(const hsk_user_lazy_args& a, int row, int b, int n,
                                                    const float* __restrict__ psrc, const float* __restrict__ msrc,
                                                    const float* __restrict__ vsrc, float* __restrict__ prow,
                                                    float* __restrict__ mrow, float* __restrict__ vrow, int lane) {
  const int B = a.B, D = a.D;
  int dl[HSK_DUP_MAX];
  int nd_list = 0;
  bool scan = false;
  if (n > 1) {
    scan = true;
    if (a.dupcnt) {
      const int nd = hsk_uniform_i(a.dupcnt[b]);
      if (lane == 0) a.dupcnt[b] = 0;
      if (nd == n - 1 && nd <= HSK_DUP_MAX) {
        const int mine = (lane < nd) ? a.duplist[b * HSK_DUP_MAX + lane] : 0x7fffffff;
        int rank = 0;
#pragma unroll
        for (int k = 0; k < HSK_DUP_MAX; ++k) rank += (k < nd && hsk_readlane_i(mine, k) < mine) ? 1 : 0;
#pragma unroll
        for (int r = 0; r < HSK_DUP_MAX; ++r) {
          const unsigned long long mm = __ballot(lane < nd && rank == r);
          dl[r] = (r < nd) ? hsk_readlane_i(mine, __builtin_ctzll(mm | (1ull << 63))) : 0;
        }
        nd_list = nd;
        scan = false;
      }
    }
  }
#pragma unroll
  for (int cc = 0; cc < NCH; ++cc) {
    const int off = (cc * 64 + lane) * V;
    const bool live = FULL || off < D;
    hsk_vec<V> p = hsk_zero<V>(), m = hsk_zero<V>(), v = hsk_zero<V>(), g = hsk_zero<V>();
    if (live) {
      p = hsk_ldg<V>(psrc + off);
      m = hsk_ldg<V>(msrc + off);
      v = hsk_ldg<V>(vsrc + off);
      if (n > 0) g = PART ? hsk_grad_chunk<V>(a, b, off) : hsk_ldg<V>(a.dUb + (long long)b * D + off);
    }
#pragma unroll
    for (int r = 0; r < HSK_DUP_MAX; ++r)
      if (r < nd_list && live) {
        const hsk_vec<V> t = PART ? hsk_grad_chunk<V>(a, dl[r], off) : hsk_ldg<V>(a.dUb + (long long)dl[r] * D + off);
#pragma unroll
        for (int q = 0; q < V; ++q) g.v[q] += t.v[q];
      }
    if (scan) {
      int found = 1;
      for (int g0 = (b / 64) * 64; g0 < B && found < n; g0 += 64) {
        const int bb = g0 + lane;
        unsigned long long mask = __ballot(bb < B && bb > b && a.u32[bb] == row);
        while (mask) {
          const int j = __builtin_ctzll(mask);
          mask &= mask - 1;
          if (live) {
            const hsk_vec<V> t = PART ? hsk_grad_chunk<V>(a, g0 + j, off) : hsk_ldg<V>(a.dUb + (long long)(g0 + j) * D + off);
#pragma unroll
            for (int q = 0; q < V; ++q) g.v[q] += t.v[q];
          }
          ++found;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < V; ++q) hsk_adamw_update<GEN>(p.v[q], m.v[q], v.v[q], g.v[q], a.c);
    if (live) {
      hsk_stg<V>(prow + off, p);
      hsk_stg<V>(mrow + off, m);
      hsk_stg<V>(vrow + off, v);
    }
  }
}

template <int V, int NCH, bool FULL, bool GEN, bool PART = false>
__device__ __forceinline__ void hsk_user_update_lazy_body(const hsk_user_lazy_args& a0, int bid) {
  hsk_user_lazy_args a = a0;
  hsk_resolve_step(a.desc, a.rel, a.ctab, a.ctab_len, a.step, a.c);
  const int B = a.B, D = a.D, step = a.step;
  if (a.fin.loss_b && bid == (B + 3) / 4) {
    hsk_finish_block(a.fin, a.c);
    return;
  }
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int b = bid * 4 + wave;
  if (b >= B) return;
  const int row = hsk_uniform_i(a.u32[b]);
  if (row < 0) return;  // empty exchange slot (row-sharded mode)
  const int own = hsk_uniform_i(a.owner[row]);
  if (own != b) return;   // a duplicate entry of a user somebody else owns
  const int n = hsk_uniform_i(a.cnt[row]);
  float* prow = a.Uw + (long long)row * D;
  float* mrow = a.mU + (long long)row * D;
  float* vrow = a.vU + (long long)row * D;
  // tab != NULL: a row that still carries pending zero-gradient steps was replayed by the forward, which left the
  // current (p, m, v) of the owner entry in ucur / mcur / vcur[b] (the table row itself is rewritten only here)
  const int done = a.tab ? hsk_uniform_i(a.last_step[row]) : step - 1;
  const bool behind = done < step - 1;
  const float* psrc = behind ? a.ucur + (long long)b * D : prow;
  const float* msrc = behind ? a.mcur + (long long)b * D : mrow;
  const float* vsrc = behind ? a.vcur + (long long)b * D : vrow;
  hsk_user_row_chunks<V, NCH, FULL, GEN, PART>(a, row, b, n, psrc, msrc, vsrc, prow, mrow, vrow, lane);
  if (lane == 0) {
    if (a.Ub) {
      float pb = a.Ub[row], mb = a.mUb[row], vb = a.vUb[row];
      for (int t = done + 1; t <= step - 1; ++t)
        hsk_adamw_update<GEN>(pb, mb, vb, 0.f, hsk_consts_at(a.c, a.tab, a.tab_len, t));
      hsk_adamw_update<GEN>(pb, mb, vb, 0.f, a.c);
      a.Ub[row] = pb;
      a.mUb[row] = mb;
      a.vUb[row] = vb;
    }
    a.last_step[row] = step;
    // last reader of the owner map this step; a duplicate entry that reads NONE afterwards just exits
    a.owner[row] = HSK_OWNER_NONE;
    a.cnt[row] = 0;
  }
}

template <int V, int NCH, bool FULL, bool GEN, bool PART = false>
__global__ __launch_bounds__(256) void k_user_update_lazy(hsk_user_lazy_args a, const int* __restrict__ g_ovf = nullptr,
                                                          const int* __restrict__ g_poison = nullptr) {
  if (hsk_guard_skip(g_ovf, g_poison)) return;
  hsk_user_update_lazy_body<V, NCH, FULL, GEN, PART>(a, (int)blockIdx.x);
}

// Dense mode (small user tables: the sweep is cheaper than replaying): AdamW on EVERY row of the table, the batch
// rows with their gradient (owner map, duplicates by batch scan), all others with g = 0 -- torch.optim's own order of
// operations.  One wave per table row; workgroup ceil(n_users/4) runs hsk_finish_block.
template <int V, int NCH, bool FULL, bool GEN, bool PART = false>
__device__ __forceinline__ void hsk_user_update_dense_body(const hsk_user_lazy_args& a0, int bid) {
  hsk_user_lazy_args a = a0;
  hsk_resolve_step(a.desc, a.rel, a.ctab, a.ctab_len, a.step, a.c);
  const int U = a.n_users, D = a.D;
  if (a.fin.loss_b && bid == (U + 3) / 4) {
    hsk_finish_block(a.fin, a.c);
    return;
  }
  const int lane = hsk_lane();
  const int row = bid * 4 + hsk_uniform_i(threadIdx.x >> 6);
  if (row >= U) return;
  float* prow = a.Uw + (long long)row * D;
  float* mrow = a.mU + (long long)row * D;
  float* vrow = a.vU + (long long)row * D;
  const int n = hsk_uniform_i(a.cnt[row]);
  const int b0 = n > 0 ? hsk_uniform_i(a.owner[row]) : 0;
  hsk_user_lazy_args a2 = a;
  a2.dupcnt = nullptr;   // the dense path never registered duplicates: batch scan
  hsk_user_row_chunks<V, NCH, FULL, GEN, PART>(a2, row, b0, n, prow, mrow, vrow, prow, mrow, vrow, lane);
  if (n > 0 && lane == 0) {
    a.owner[row] = HSK_OWNER_NONE;
    a.cnt[row] = 0;
  }
  if (a.Ub && lane == 0) {
    // d loss / d user_bias is identically 0 under BPR (it cancels in s_pos - s_neg)
    float pb = a.Ub[row], mb = a.mUb[row], vb = a.vUb[row];
    hsk_adamw_update<GEN>(pb, mb, vb, 0.f, a.c);
    a.Ub[row] = pb;
    a.mUb[row] = mb;
    a.vUb[row] = vb;
  }
}

template <int V, int NCH, bool FULL, bool GEN, bool PART = false>
__global__ __launch_bounds__(256) void k_user_update_dense(hsk_user_lazy_args a) {
  hsk_user_update_dense_body<V, NCH, FULL, GEN, PART>(a, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Catch-up AHEAD of time.  The replay of a row's pending zero-gradient steps is pure VALU work (two transcendentals
// per element and step); inside the forward it sits at the head of every wave's critical path.  The step knows its
// NEXT batch (the prefetch hint), so workgroups riding in the item pass's launch -- memory-bound, VALU idle -- bring
// the next batch's user rows up to the CURRENT step (inclusive: those users are not in the current batch, so the
// current step is one more zero-gradient step for them) and the next forward finds them current.  A row that is in
// the current batch too is skipped (`stamp`): its owner is rewriting it in this very launch.  Two entries of the next
// batch naming the same user: the first to raise `claim[u]` to this step does the work.  Purely an optimisation: a row
// nobody brought up to date is replayed by the forward as before; results are bit-identical either way.
// ---------------------------------------------------------------------------------------------
struct hsk_ahead_args {
  const int32_t* coo_user;   // NULL: nothing to do
  const int64_t* order;
  long long start;           // next batch: interactions order[start .. start + n)
  int n;
  const int* stamp_cur;      // stamp of the batch being trained on (== step for its users)
  int* claim;
  float* Uw; float* mU; float* vU; float* Ub; float* mUb; float* vUb;
  int* last_step;
  int D, step;               // step: the step being taken
  hsk_adamw_consts c;
  const float2* tab; int tab_len;
  const hsk_step_desc* desc; int rel;   // graph replay: next batch = desc batch rel + 1
};

// One WAVE per entry of the next batch (4 per workgroup): a wave slot is what these workgroups take from the item pass.
template <int V, int NCH, bool FULL, bool GEN>
__device__ __forceinline__ void hsk_user_ahead_body(const hsk_ahead_args& a, int bid) {
  const int lane = hsk_lane();
  const int b = bid * 4 + hsk_uniform_i(threadIdx.x >> 6);
  if (b >= a.n) return;
  int step = a.step;
  long long start = a.start;
  const int64_t* order = a.order;
  if (a.desc) {
    step = a.desc->step0 + a.rel + 1;
    start = a.desc->start0 + (long long)(a.rel + 1) * a.n;
    order = a.desc->order;
  }
  const long long pos = order ? (long long)order[start + b] : (start + b);
  const int u = hsk_uniform_i(a.coo_user[pos]);
  if (hsk_uniform_i(a.stamp_cur[u]) == step) return;   // in the current batch: being updated right now
  const int done = hsk_uniform_i(a.last_step[u]);
  if (done >= step) return;
  int won = 0;
  if (lane == 0) won = atomicMax(&a.claim[u], step) < step;
  if (!hsk_uniform_i(won)) return;                     // another entry of the next batch names the same user
  const int D = a.D;
  float* prow = a.Uw + (long long)u * D;
  float* mrow = a.mU + (long long)u * D;
  float* vrow = a.vU + (long long)u * D;
  hsk_row<V, NCH> p, m, v;
  hsk_row_load<V, NCH, FULL>(p, prow, lane, D);
  hsk_row_load<V, NCH, FULL>(m, mrow, lane, D);
  hsk_row_load<V, NCH, FULL>(v, vrow, lane, D);
  for (int t0 = done + 1; t0 <= step; t0 += 64) {
    const float2 blk = hsk_consts_block(a.tab, a.tab_len, t0);
    const int nt = min(64, step - t0 + 1);
    for (int j = 0; j < nt; ++j) {
      const hsk_adamw_consts ct = hsk_consts_lane(a.c, blk, j);
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
        for (int q = 0; q < V; ++q) hsk_adamw_replay<GEN>(p.c[cc].v[q], m.c[cc].v[q], v.c[cc].v[q], ct);
    }
  }
  hsk_row_store<V, NCH, FULL>(p, prow, lane, D);
  hsk_row_store<V, NCH, FULL>(m, mrow, lane, D);
  hsk_row_store<V, NCH, FULL>(v, vrow, lane, D);
  if (lane == 0) {
    if (a.Ub) {
      float pb = a.Ub[u], mb = a.mUb[u], vb = a.vUb[u];
      for (int t = done + 1; t <= step; ++t) hsk_adamw_update<GEN>(pb, mb, vb, 0.f, hsk_consts_at(a.c, a.tab, a.tab_len, t));
      a.Ub[u] = pb;
      a.mUb[u] = mb;
      a.vUb[u] = vb;
    }
    a.last_step[u] = step;
  }
}

// bring every row with last_step < step up to `step` (one workgroup per table row)
template <int VV, bool GEN>
__global__ __launch_bounds__(256) void k_user_flush(float* __restrict__ Uw, float* __restrict__ mU,
                                                    float* __restrict__ vU, float* __restrict__ Ub,
                                                    float* __restrict__ mUb, float* __restrict__ vUb,
                                                    int* __restrict__ last_step, int n_users, int D, int step,
                                                    hsk_adamw_consts c, const float2* __restrict__ tab, int tab_len,
                                                    const int* __restrict__ g_poison = nullptr) {
  if (hsk_guard_skip(nullptr, g_poison)) return;
  const int row = blockIdx.x;
  const int done = last_step[row];
  if (done >= step) return;
  hsk_row_replay_wg<VV, GEN>(Uw + (long long)row * D, mU + (long long)row * D, vU + (long long)row * D, D, done, step, c,
                             tab, tab_len);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (Ub) {
      float pb = Ub[row], mb = mUb[row], vb = vUb[row];
      for (int t = done + 1; t <= step; ++t) hsk_adamw_update<GEN>(pb, mb, vb, 0.f, hsk_consts_at(c, tab, tab_len, t));
      Ub[row] = pb;
      mUb[row] = mb;
      vUb[row] = vb;
    }
    last_step[row] = step;
  }
}

// The same sweep with one WAVE per table row (lane l holds (c*64 + l)*V .. +V of chunk c, hsk_rows.h): 16-byte accesses
// and all three operands of a row in flight per wave -- at D = 512 a CU keeps 32 rows = 192 KB outstanding against the
// 8 x 6 KB of the workgroup-per-row form, which is what a sweep bounded by HBM latency x bytes in flight wants.  Every
// element sees the operations of k_user_flush in the same order (hsk_adamw_replay over the same constants): same bits.
template <int V, int NCH, bool FULL, bool GEN>
__global__ __launch_bounds__(256) void k_row_flush_wave(float* __restrict__ Uw, float* __restrict__ mU,
                                                        float* __restrict__ vU, float* __restrict__ Ub,
                                                        float* __restrict__ mUb, float* __restrict__ vUb,
                                                        int* __restrict__ last_step, int n_rows, int D, int step,
                                                        hsk_adamw_consts c, const float2* __restrict__ tab, int tab_len,
                                                        const int* __restrict__ g_poison = nullptr) {
  if (hsk_guard_skip(nullptr, g_poison)) return;
  const int lane = hsk_lane();
  const int row = blockIdx.x * 4 + hsk_uniform_i(threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int done = hsk_uniform_i(last_step[row]);
  if (done >= step) return;
  float* prow = Uw + (long long)row * D;
  float* mrow = mU + (long long)row * D;
  float* vrow = vU + (long long)row * D;
  hsk_row<V, NCH> p, m, v;
  hsk_row_load<V, NCH, FULL>(p, prow, lane, D);
  hsk_row_load<V, NCH, FULL>(m, mrow, lane, D);
  hsk_row_load<V, NCH, FULL>(v, vrow, lane, D);
  for (int t0 = done + 1; t0 <= step; t0 += 64) {
    const float2 blk = hsk_consts_block(tab, tab_len, t0);
    const int nt = min(64, step - t0 + 1);
    for (int j = 0; j < nt; ++j) {
      const hsk_adamw_consts ct = hsk_consts_lane(c, blk, j);
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
        for (int q = 0; q < V; ++q) hsk_adamw_replay<GEN>(p.c[cc].v[q], m.c[cc].v[q], v.c[cc].v[q], ct);
    }
  }
  hsk_row_store<V, NCH, FULL>(p, prow, lane, D);
  hsk_row_store<V, NCH, FULL>(m, mrow, lane, D);
  hsk_row_store<V, NCH, FULL>(v, vrow, lane, D);
  if (lane == 0) {
    if (Ub) {
      float pb = Ub[row], mb = mUb[row], vb = vUb[row];
      for (int t = done + 1; t <= step; ++t) hsk_adamw_update<GEN>(pb, mb, vb, 0.f, hsk_consts_at(c, tab, tab_len, t));
      Ub[row] = pb;
      mUb[row] = mb;
      vUb[row] = vb;
    }
    last_step[row] = step;
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

// n_part > 1: it32 rows have n_cols + n_part - 1 columns, the positive in the first n_part (k_prep_sample)
__global__ __launch_bounds__(256) void k_widen_batch(const int* __restrict__ u32, const int* __restrict__ it32,
                                                     long long B, long long total, int64_t* __restrict__ u_out,
                                                     int64_t* __restrict__ i_out, int n_cols = 0, int n_part = 1) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) {
    if (n_part > 1) {
      const long long b = e / n_cols, k = e - b * n_cols;
      i_out[e] = it32[b * (n_cols + n_part - 1) + (k == 0 ? 0 : k + n_part - 1)];
    } else {
      i_out[e] = it32[e];
    }
  }
  if (e < B) u_out[e] = u32[e];
}

// undo the owner-map contribution of a batch that will not be trained on (a discarded prefetch)
__global__ __launch_bounds__(256) void k_release_owner(const int* __restrict__ u32, int B, int* __restrict__ owner,
                                                       int* __restrict__ cnt) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int u = u32[b];
  owner[u] = HSK_OWNER_NONE;
  cnt[u] = 0;
}
