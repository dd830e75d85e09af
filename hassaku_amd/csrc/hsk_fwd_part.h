// hsk_fwd_part.h -- the forward of the fused step with the item table cut into P range partitions, one partition per
// group of 8/P XCDs.
//
// k_fwd_ugrad gathers uniformly random rows of the whole item table on every XCD: a 4 MB L2 holds 18 % of a 21.9 MB
// table, the rest of the 850 MB per step comes over the fabric from the Infinity Cache (8.3 TB/s, 104 us).  Here a wave
// owns (positive b, partition q): it weighs only the negatives of b whose item lies in partition q, and the workgroups
// of partition q all run on the same 8/P XCDs (workgroup index mod 8 -> XCD, the observed round-robin dispatch: a speed
// assumption only), so an XCD's L2 only ever sees I/P item rows -- at P = 4 and the ml10m shape 5.5 MB, three quarters
// of it resident.  Measured stand-alone (profiles/probes/part_fwd.hip, us): P = 1 106 | 2 89 | 4 72 | 8 78.
//
// Nothing is exchanged inside the launch.  Every unit loads the user row and the positive's item row itself (s_0 is
// recomputed by each of the P units from the same bytes in the same order: identical bits), picks ITS negatives out of
// the positive's row of ids (one ballot per 64 columns, a list in LDS), weighs them and leaves PARTIAL results: a
// partial user-row gradient dUp[q][b] (its share of the positive's term included: -gsum_q * I[i_0]), its share -gsum_q
// of d loss/d s_0, and a partial loss term.  The batch rows have K + P - 1 columns: the positive item, P - 1 columns
// that are no entries (-1: the item sort skips them), then the negatives (k_prep_sample); unit q leaves its share of
// d loss/d s_0 in column q of g_s, and the item pass adds the P shares when it meets the positive's entry
// (hsk_entry_weight).  The owner's user update adds the P partial rows (hsk_user_row_chunks<PART>).
//
// Lazy user rows: the ahead-of-time catch-up (hsk_user_ahead_body, workgroups of this launch for the NEXT batch) keeps
// the rows current; a row that is behind all the same (no hint, first step) is replayed in registers by each of its P
// units, unit 0 publishes it -- correct, just P times the arithmetic for that row.
#pragma once
#include "hsk_sampler.h"
#include "hsk_sort.h"
#include "hsk_step_kernels.h"

// In-launch preparation pipeline (hsk_fused.hip: hsk_pipe_*): phases of the batches one and two steps ahead ride in
// the step's two launches as extra workgroups at the head of the grid, instead of five launches on a side stream --
// no cross-stream events on the step's path, and the launch boundaries the step has anyway are the phases' barriers.
//   forward of step t     bucket phase of batch t, sampler of batch t+2, row scan of batch t+1
//   item/user launch of t scatter of batch t+1, histogram of batch t+2
// n_total = the riding workgroups padded to a multiple of 8 (the workgroups behind keep index % 8 == their own index % 8).
struct hsk_ride_fwd {
  int n_total;
  hsk_ride_sort Bk;
  hsk_ride_sample S;
  hsk_ride_sort R;
};
struct hsk_ride_item {
  int n_total;
  hsk_ride_sort C;
  hsk_ride_sort H;
};

struct hsk_part_args {
  int n_part;                    // P: 2, 4 or 8; partition q = items [ceil(q I / P), ceil((q + 1) I / P))
  int n_items;
  int n_ahead_blocks;            // workgroups that run hsk_user_ahead_body (a multiple of 8)
  int ahead_stride;              // > 0: one OCTET of ahead workgroups follows every `ahead_stride` octets of unit
                                 // workgroups (octets: a unit workgroup keeps blockIdx % 8 == its own index % 8), the
                                 // replay -- VALU work -- runs beside the gathers all along; 0: the ahead workgroups lead
};
#define HSK_PART_LIST_MAX 256    // negatives per positive (the partition rule keeps n_neg <= 256)

// The riding phases of the forward's launch.  (Any of them in the kernel costs the gather loop's register allocation one
// register: 129 with four item rows per buffer -- three waves per SIMD; capped at 128 it spills and the kernel runs 86 ->
// 96 us; called out of line it needs a stack.  Hence HSK_FWD_PART_R = 3 rows per buffer: 114 registers.)
__device__ __forceinline__ void hsk_ride_fwd_roles(const hsk_ride_fwd& rf, int bid) {
  if (bid < rf.Bk.n_blocks) {
    __shared__ int bucket_lds[5 * HSK_PIPE_MAX_IPB + 2];
    hsk_sort_bucket_body(rf.Bk.perm1, rf.Bk.n_entries, rf.Bk.n_items, rf.Bk.plan, rf.Bk.bstart, rf.Bk.perm, rf.Bk.offsets,
                         nullptr, nullptr, bid, bucket_lds);
    return;
  }
  bid -= rf.Bk.n_blocks;
  if (bid < rf.S.n_blocks) {
    hsk_ride_sample_body(rf.S, bid);
    return;
  }
  bid -= rf.S.n_blocks;
  if (bid < rf.R.n_blocks) hsk_sort_rowscan_body(rf.R.hist, rf.R.plan, rf.R.btot, bid);
}

template <int V, int NCH, bool FULL, int R, int LOSS, bool GEN>
__global__ __launch_bounds__(256) void k_fwd_part(const float* __restrict__ Uw, const float* __restrict__ Iw,
                                                  const float* __restrict__ Ib, const int* __restrict__ u32,
                                                  const int* __restrict__ it32, int B, int K /* columns */, int D, float inv_norm,
                                                  float* __restrict__ g_s, float* __restrict__ dUp,
                                                  double* __restrict__ loss_p, hsk_lazy_user_args lz, hsk_part_args pa,
                                                  hsk_ahead_args aa, hsk_ride_fwd rf) {
  static_assert(LOSS == HSK_LOSS_BPR || LOSS == HSK_LOSS_BCE, "the sampled softmax needs all negatives in one wave");
  int bid = (int)blockIdx.x;
  if (bid < rf.n_total) {   // preparation phases of later batches (see hsk_ride_fwd)
    hsk_ride_fwd_roles(rf, bid);
    return;
  }
  bid -= rf.n_total;
  if (pa.ahead_stride > 0) {
    const int o = bid >> 3, r = bid & 7, n_ao = pa.n_ahead_blocks >> 3;
    const int period = pa.ahead_stride + 1, full = n_ao * period;
    if (o < full) {
      const int k = o / period, j = o - k * period;
      if (j == pa.ahead_stride) {
        hsk_user_ahead_body<V, NCH, FULL, GEN>(aa, k * 8 + r);
        return;
      }
      bid = (k * pa.ahead_stride + j) * 8 + r;
    } else {
      bid = (n_ao * pa.ahead_stride + (o - full)) * 8 + r;
    }
  } else {
    if (bid < pa.n_ahead_blocks) {
      hsk_user_ahead_body<V, NCH, FULL, GEN>(aa, bid);
      return;
    }
    bid -= pa.n_ahead_blocks;
  }
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int P = pa.n_part, M = 8 / P;
  const int xcd = bid & 7, q = xcd / M, sx = xcd - q * M, t = bid >> 3;
  const int b = (t * M + sx) * 4 + wave;
  if (b >= B) return;

  using Row = hsk_row<V, NCH>;
  const int* __restrict__ irow = it32 + (long long)b * K;
  // Loads in three rounds instead of a chain of six: (1) what depends on b only -- the user id, the positive, the row of
  // item ids; (2) the user row, the positive's row, the user's lazy-update state and the first item rows; (3) the rest.
  const int u_v = u32[b];
  const int i0_v = irow[0];
  int idv[HSK_PART_LIST_MAX / 64];
#pragma unroll
  for (int r = 0; r < HSK_PART_LIST_MAX / 64; ++r) {
    const int c = P + r * 64 + lane;
    idv[r] = (c < K) ? irow[c] : -1;
  }
  const int u = hsk_uniform_i(u_v);
  const int i0 = hsk_uniform_i(i0_v);
  Row ur, r0, acc;
  hsk_row_load<V, NCH, FULL>(ur, Uw + (long long)u * D, lane, D);
  hsk_row_load<V, NCH, FULL>(r0, Iw + (long long)i0 * D, lane, D);
  int done_v = 0, own_v = 0;
  if (lz.mU) {
    done_v = lz.last_step[u];
    own_v = lz.owner[u];
  }
  const float bias0 = Ib ? Ib[i0] : 0.f;
  // this unit's negatives: (item id, column) of the ids in [lo, hi), in column order
  __shared__ int lst_id[4][HSK_PART_LIST_MAX], lst_col[4][HSK_PART_LIST_MAX];
  const int lo = (int)(((long long)q * pa.n_items + P - 1) / P), hi = (int)(((long long)(q + 1) * pa.n_items + P - 1) / P);
  int n_mine = 0;
#pragma unroll
  for (int r = 0; r < HSK_PART_LIST_MAX / 64; ++r) {
    const bool mine = idv[r] >= lo && idv[r] < hi;
    const unsigned long long m = __ballot(mine);
    if (mine) {
      const int pos = n_mine + __popcll(m & ((1ull << lane) - 1ull));
      lst_id[wave][pos] = idv[r];
      lst_col[wave][pos] = P + r * 64 + lane;
    }
    n_mine += __popcll(m);
  }
  __builtin_amdgcn_wave_barrier();

  int kc = 0;
  int nr = min(64, n_mine);
  int myidx = (lane < nr) ? lst_id[wave][lane] : i0;
  int mycol = (lane < nr) ? lst_col[wave][lane] : 0;
  Row bufA[R], bufB[R];
  auto prefetch = [&](Row(&buf)[R], int j) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (j + r < nr)
        hsk_row_load<V, NCH, FULL>(buf[r], Iw + (long long)hsk_readlane_i(myidx, j + r) * D, lane, D);
  };
  prefetch(bufA, 0);
  float mybias = Ib ? Ib[myidx] : 0.f;

  if (lz.mU)
    hsk_user_row_current<V, NCH, FULL>(ur, u, b, B, D, lane, lz, q == 0, hsk_uniform_i(done_v), hsk_uniform_i(own_v));
  else if (lz.ucur && q == 0)
    hsk_row_store<V, NCH, FULL>(ur, lz.ucur + (long long)b * D, lane, D);
  hsk_row_zero(acc);
  const float s0 = hsk_wave_sum(hsk_row_dot_partial(ur, r0)) + bias0;

  float gsum = 0.f;
  double lsum = 0.0;
  while (true) {
    float gv = 0.f, xv = 0.f;
    auto process = [&](Row(&buf)[R], int j) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (j + r < nr) {
          const float s = hsk_wave_sum(hsk_row_dot_partial(ur, buf[r])) + hsk_readlane_f(mybias, j + r);
          float g, x;
          if (LOSS == HSK_LOSS_BPR) {
            x = s0 - s;
            g = inv_norm / (1.f + expf(x));    // sigma(-x)/(B*N) = d loss / d s_neg
            gsum += g;
          } else {
            x = s;
            g = inv_norm / (1.f + expf(-s));   // sigma(s)/(B*K), label 0
          }
          hsk_row_axpy(acc, g, buf[r]);
          gv = (lane == j + r) ? g : gv;
          xv = (lane == j + r) ? x : xv;
        }
      }
    };
    for (int j = 0; j < nr; j += 2 * R) {
      prefetch(bufB, j + R);
      process(bufA, j);
      prefetch(bufA, j + 2 * R);
      process(bufB, j + R);
    }
    if (lane < nr) {
      g_s[(long long)b * K + mycol] = gv;
      lsum += (double)hsk_softplus(LOSS == HSK_LOSS_BPR ? -xv : xv);
    }
    kc += 64;
    if (kc >= n_mine) break;
    nr = min(64, n_mine - kc);
    myidx = (lane < nr) ? lst_id[wave][kc + lane] : i0;
    mycol = (lane < nr) ? lst_col[wave][kc + lane] : 0;
    prefetch(bufA, 0);
    mybias = Ib ? Ib[myidx] : 0.f;
  }
  float gp;   // this unit's share of -d loss/d s_0
  if (LOSS == HSK_LOSS_BPR) {
    gp = gsum;
  } else {
    gp = (q == 0) ? inv_norm / (1.f + expf(s0)) : 0.f;   // -(sigma(s_0) - 1)/(B*K), once per positive
    if (q == 0 && lane == 0) lsum += (double)hsk_softplus(-s0);
  }
  hsk_row_axpy(acc, -gp, r0);
  hsk_row_store<V, NCH, FULL>(acc, dUp + ((long long)b * P + q) * D, lane, D);   // [b][q][D]
  const double l = hsk_wave_sum_f64(lsum);
  if (lane == 0) {
    g_s[(long long)b * K + q] = -gp;
    loss_p[(long long)q * B + b] = l;
  }
}
