// hsk_fwd_small.h -- forward kernel for small batches: one WORKGROUP per positive, its (1+N) columns split over the
// four waves.
//
// k_fwd_ugrad gives a positive to one wave, which walks its 1+N item rows with 8 in flight: at B = 4096 that is
// 4096 waves = every wave slot of the chip, and the walk is bandwidth-bound.  At the reference's usual batch sizes
// (128..512) it is 128..512 waves on 1024 SIMDs and the walk is a chain of ~N/8 memory latencies: 24 us at
// B = 128, N = 50.  Here each of the 4 waves takes a quarter of the columns (and recomputes s_0 itself); the partial
// user-row gradients and sums meet in LDS and wave 0 adds them in wave order -- deterministic, but a different
// summation order than k_fwd_ugrad's.  BPR and BCE (the sampled softmax keeps the one-wave kernel: its running
// max / normaliser would need a second combine).
#pragma once
#include "hsk_rows.h"

// NW waves per workgroup (4 or 8): with 8, a positive's 50 negatives are 6-7 rows per wave -- all in flight at once,
// i.e. one round of memory latency instead of two.
template <int V, int NCH, bool FULL, int R, int LOSS, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k_fwd_ugrad_wg(const float* __restrict__ Uw, const float* __restrict__ Iw,
                                                      const float* __restrict__ Ib, const int* __restrict__ u32,
                                                      const int* __restrict__ it32, int B, int K, int D,
                                                      float inv_norm, float* __restrict__ g_s,
                                                      float* __restrict__ dUb, double* __restrict__ loss_b,
                                                      hsk_lazy_user_args lz = hsk_lazy_user_args{}) {
  extern __shared__ float lds_rows[];   // [NW][D]: partial gradient rows of waves 1.. (slot 0 unused)
  __shared__ float sh_gsum[NW];
  __shared__ double sh_loss[NW];
  const int lane = hsk_lane();
  const int sub = hsk_uniform_i(threadIdx.x >> 6);
  const int b = blockIdx.x;
  if (b >= B) return;

  using Row = hsk_row<V, NCH>;
  const int* __restrict__ irow = it32 + (long long)b * K;
  // Loads in two rounds, not a chain of four (a small step is a chain of latencies): (1) what depends on b only -- user
  // id, positive, this wave's first 64 item ids; (2) user row, positive's row, lazy-update state, biases, first item rows.
  const int per = (K - 1 + NW - 1) / NW;           // columns per wave
  const int k_lo = 1 + sub * per, k_hi = min(K, k_lo + per);
  int kc = k_lo;
  int nr = max(0, min(64, k_hi - kc));
  const int u_v = u32[b];
  const int i0_v = irow[0];
  const int id_v = (lane < nr) ? irow[kc + lane] : -1;
  const int u = hsk_uniform_i(u_v);
  const int i0 = hsk_uniform_i(i0_v);
  int myidx = (lane < nr) ? id_v : i0;
  Row ur, r0, acc;
  hsk_row_load<V, NCH, FULL>(ur, Uw + (long long)u * D, lane, D);
  hsk_row_load<V, NCH, FULL>(r0, Iw + (long long)i0 * D, lane, D);
  int done_v = 0, own_v = 0;
  if (lz.mU) {
    done_v = lz.last_step[u];
    own_v = lz.owner[u];
  }
  const float bias0 = Ib ? Ib[i0] : 0.f;
  float mybias = Ib ? Ib[myidx] : 0.f;
  Row bufA[R], bufB[R];
  auto prefetch = [&](Row(&buf)[R], int j) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (j + r < nr) hsk_row_load<V, NCH, FULL>(buf[r], Iw + (long long)hsk_readlane_i(myidx, j + r) * D, lane, D);
  };
  prefetch(bufA, 0);
  // lazily updated user row: every wave replays the pending steps on its own copy (identical results), wave 0
  // publishes the current row for the item pass
  if (lz.mU)
    hsk_user_row_current<V, NCH, FULL>(ur, u, b, B, D, lane, lz, sub == 0, hsk_uniform_i(done_v), hsk_uniform_i(own_v));
  else if (lz.ucur && sub == 0)
    hsk_row_store<V, NCH, FULL>(ur, lz.ucur + (long long)b * D, lane, D);
  hsk_row_zero(acc);
  const float s0 = hsk_wave_sum(hsk_row_dot_partial(ur, r0)) + bias0;

  float gsum = 0.f;
  double lsum = 0.0;
  while (nr > 0) {
    float gv = 0.f, xv = 0.f;
    auto process = [&](Row(&buf)[R], int j) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (j + r < nr) {
          const float s = hsk_wave_sum(hsk_row_dot_partial(ur, buf[r])) + hsk_readlane_f(mybias, j + r);
          float g;
          if (LOSS == HSK_LOSS_BPR) {
            const float x = s0 - s;
            g = inv_norm / (1.f + expf(x));     // sigma(-x)/(B*N)
            gsum += g;
            xv = (lane == j + r) ? x : xv;
          } else {
            g = inv_norm / (1.f + expf(-s));    // sigma(s)/(B*K), label 0
            xv = (lane == j + r) ? s : xv;
          }
          hsk_row_axpy(acc, g, buf[r]);
          gv = (lane == j + r) ? g : gv;
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
      g_s[(long long)b * K + kc + lane] = gv;
      lsum += (double)(LOSS == HSK_LOSS_BPR ? hsk_softplus(-xv) : hsk_softplus(xv));
    }
    kc += 64;
    nr = max(0, min(64, k_hi - kc));
    if (nr == 0) break;
    myidx = (lane < nr) ? irow[kc + lane] : i0;
    mybias = Ib ? Ib[myidx] : 0.f;
    prefetch(bufA, 0);
  }

  // combine: waves 1..3 park their partial row / sums in LDS, wave 0 adds them in wave order
  const double l = hsk_wave_sum_f64(lsum);
  if (sub > 0) hsk_row_store<V, NCH, FULL>(acc, lds_rows + (long long)sub * D, lane, D);
  if (lane == 0) {
    sh_gsum[sub] = gsum;
    sh_loss[sub] = l;
  }
  __syncthreads();
  if (sub != 0) return;
#pragma unroll
  for (int w = 1; w < NW; ++w) {
    Row t;
    hsk_row_load<V, NCH, FULL>(t, lds_rows + (long long)w * D, lane, D);
    hsk_row_add(acc, t);
  }
  float g0;
  double ltot = sh_loss[0];
  float gtot = sh_gsum[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) {   // wave order: deterministic
    ltot += sh_loss[w];
    gtot += sh_gsum[w];
  }
  if (LOSS == HSK_LOSS_BPR) {
    g0 = -gtot;   // d loss / d s_pos
  } else {
    g0 = -inv_norm / (1.f + expf(s0));   // (sigma(s_0) - 1)/(B*K)
    ltot += (double)hsk_softplus(-s0);
  }
  hsk_row_axpy(acc, g0, r0);
  if (lane == 0) {
    g_s[(long long)b * K] = g0;
    loss_b[b] = ltot;
  }
  hsk_row_store<V, NCH, FULL>(acc, dUb + (long long)b * D, lane, D);
}
