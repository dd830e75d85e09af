// hsk_ops.hip -- un-fused BPR-MF operators for gfx950 (MI355X): sampler, gather+score+loss+row-grads,
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
#include "hsk_rows.h"

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

// dim == 0: the bias-only model (SGDBaseline, algorithms/sgd_alg.py:72-107): out = ub[u] + ib[i] + gb, one thread per logit
__global__ __launch_bounds__(256) void k_bias_scores(const float* __restrict__ Ib, const float* __restrict__ Ub,
                                                     const float* __restrict__ gb, int n_users, int n_items,
                                                     const int64_t* __restrict__ u_idx,
                                                     const int64_t* __restrict__ i_idx, long long B, long long K,
                                                     float* __restrict__ out, int32_t* status) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * K) return;
  const int u = hsk_clamp_index(u_idx[e / K], n_users, status);
  const int it = hsk_clamp_index(i_idx[e], n_items, status);
  // the reference's order: u_repr + i_repr + global_bias
  float s = Ub ? Ub[u] : 0.f;
  s = s + (Ib ? Ib[it] : 0.f);
  if (gb) s = s + gb[0];
  out[e] = s;
}

// gradients of the bias-only model: one wave per row b (sums its K logit gradients for ub[u_b] and gb)
__global__ __launch_bounds__(256) void k_bias_backward(int n_users, int n_items, const int64_t* __restrict__ u_idx,
                                                       const int64_t* __restrict__ i_idx, int B, long long K,
                                                       const float* __restrict__ gl, float* __restrict__ gIb,
                                                       float* __restrict__ gUb, float* __restrict__ ggb,
                                                       int32_t* status) {
  const int lane = hsk_lane();
  const int b = blockIdx.x * 4 + hsk_uniform_i(threadIdx.x >> 6);
  if (b >= B) return;
  float gsum = 0.f;
  for (long long k = lane; k < K; k += 64) {
    const float g = gl[(long long)b * K + k];
    gsum += g;
    if (gIb) atomicAdd(&gIb[hsk_clamp_index(i_idx[(long long)b * K + k], n_items, status)], g);
  }
  const float gs = hsk_wave_sum(gsum);
  if (lane == 0) {
    if (gUb) atomicAdd(&gUb[hsk_clamp_index(u_idx[b], n_users, status)], gs);
    if (ggb) atomicAdd(ggb, gs);
  }
}

extern "C" int hsk_mf_scores(const float* user_emb, const float* item_emb, const float* item_bias,
                             const float* user_bias, const float* global_bias, int64_t n_users, int64_t n_items,
                             int64_t dim, const int64_t* u_idx, const int64_t* i_idx, int64_t batch, int64_t n_cols,
                             float* logits, int32_t* status, hsk_stream_t stream_) {
  HSK_REQUIRE(u_idx && i_idx && logits, HSK_ERR_INVALID, "NULL pointer argument");
  HSK_REQUIRE(n_users > 0 && n_items > 0 && dim >= 0 && batch >= 0 && n_cols >= 0, HSK_ERR_INVALID, "bad sizes");
  HSK_REQUIRE(dim == 0 || (user_emb && item_emb), HSK_ERR_INVALID, "embedding tables must not be NULL when dim > 0");
  HSK_REQUIRE(batch <= 65535, HSK_ERR_UNSUPPORTED, "batch %lld > 65535 rows per call", (long long)batch);
  if (batch == 0 || n_cols == 0) return HSK_OK;
  hipStream_t stream = (hipStream_t)stream_;
  if (dim == 0) {
    k_bias_scores<<<(unsigned)hsk_ceil_div(batch * n_cols, 256), 256, 0, stream>>>(
        item_bias, user_bias, global_bias, (int)n_users, (int)n_items, u_idx, i_idx, (long long)batch, (long long)n_cols,
        logits, status);
    HSK_LAUNCH_CHECK();
    return HSK_OK;
  }
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
  HSK_REQUIRE(u_idx && i_idx && grad_logits, HSK_ERR_INVALID, "NULL pointer argument");
  HSK_REQUIRE(n_users > 0 && n_items > 0 && dim >= 0 && batch >= 0 && n_cols >= 0, HSK_ERR_INVALID, "bad sizes");
  HSK_REQUIRE(dim == 0 || (user_emb && item_emb), HSK_ERR_INVALID, "embedding tables must not be NULL when dim > 0");
  HSK_REQUIRE(batch <= 65535, HSK_ERR_UNSUPPORTED, "batch %lld > 65535 rows per call", (long long)batch);
  hipStream_t stream = (hipStream_t)stream_;
  if (dim == 0) {   // bias-only model
    if (g_item_bias) HSK_HIP(hipMemsetAsync(g_item_bias, 0, (size_t)n_items * 4, stream));
    if (g_user_bias) HSK_HIP(hipMemsetAsync(g_user_bias, 0, (size_t)n_users * 4, stream));
    if (g_global_bias) HSK_HIP(hipMemsetAsync(g_global_bias, 0, 4, stream));
    if (batch == 0 || n_cols == 0) return HSK_OK;
    k_bias_backward<<<(unsigned)hsk_ceil_div(batch, 4), 256, 0, stream>>>((int)n_users, (int)n_items, u_idx, i_idx,
                                                                          (int)batch, (long long)n_cols, grad_logits,
                                                                          g_item_bias, g_user_bias, g_global_bias, status);
    HSK_LAUNCH_CHECK();
    return HSK_OK;
  }
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

extern "C" int hsk_opt_dense(int opt_kind, float* p, const float* g, float* m, float* v, int64_t n, double lr,
                             double beta1, double beta2, double eps, double wd, int64_t step, hsk_stream_t stream_) {
  HSK_REQUIRE(opt_kind >= HSK_OPT_ADAMW && opt_kind <= HSK_OPT_ADAGRAD, HSK_ERR_INVALID, "unknown opt_kind %d", opt_kind);
  HSK_REQUIRE(p && m && v, HSK_ERR_INVALID, "NULL pointer argument");
  HSK_REQUIRE(n >= 0 && step >= 1, HSK_ERR_INVALID, "need n >= 0 and step >= 1");
  HSK_REQUIRE((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g) & 15) == 0, HSK_ERR_INVALID,
              "buffers must be 16-byte aligned");
  if (n == 0) return HSK_OK;
  const hsk_adamw_consts c = hsk_make_adamw_consts(lr, beta1, beta2, eps, wd, step, opt_kind);
  const int64_t n4 = n >> 2;
  int64_t blocks = hsk_ceil_div(n4 > 0 ? n4 : 1, 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  k_adamw_dense<<<(unsigned)blocks, 256, 0, (hipStream_t)stream_>>>(p, g, m, v, (long long)n, c);
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

extern "C" int hsk_adamw_dense(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                               double beta2, double eps, double wd, int64_t step, hsk_stream_t stream_) {
  return hsk_opt_dense(HSK_OPT_ADAMW, p, g, m, v, n, lr, beta1, beta2, eps, wd, step, stream_);
}

// bce / sampled-softmax loss + gradient wrt logits, one wave per row (see hsk_rec_loss_grad in the header)
__global__ __launch_bounds__(256) void k_rec_loss_grad(int kind, const float* __restrict__ logits, int B, long long K,
                                                       float inv_norm, float log_adjust, float* __restrict__ grad,
                                                       double* __restrict__ ws) {
  const int lane = hsk_lane();
  const int wave = hsk_uniform_i(threadIdx.x >> 6);
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const float* row = logits + (long long)b * K;
  float* grow = grad ? grad + (long long)b * K : nullptr;
  double lsum = 0.0;
  if (kind == HSK_LOSS_BCE) {
    for (long long k = lane; k < K; k += 64) {
      const float s = row[k];
      const bool pos = (k == 0);
      lsum += (double)hsk_softplus(pos ? -s : s);
      if (grow) grow[k] = pos ? -inv_norm / (1.f + expf(s)) : inv_norm / (1.f + expf(-s));
    }
  } else {  // sampled softmax
    float mx = -INFINITY;
    for (long long k = lane; k < K; k += 64) mx = fmaxf(mx, row[k] + (k == 0 ? 0.f : log_adjust));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = 0.f;
    for (long long k = lane; k < K; k += 64) sum += expf(row[k] + (k == 0 ? 0.f : log_adjust) - mx);
    sum = hsk_wave_sum(sum);
    const float rinv = inv_norm / sum;
    if (grow)
      for (long long k = lane; k < K; k += 64)
        grow[k] = expf(row[k] + (k == 0 ? 0.f : log_adjust) - mx) * rinv - (k == 0 ? inv_norm : 0.f);
    if (lane == 0) lsum = (double)(-row[0] + mx + logf(sum));
  }
  const double l = hsk_wave_sum_f64(lsum);
  if (lane == 0) ws[b] = l;
}

extern "C" int hsk_rec_loss_grad(int32_t kind, const float* logits, int64_t batch, int64_t n_cols, double log_adjust,
                                 double* loss, float* grad_logits, double* ws, hsk_stream_t stream_) {
  if (kind == HSK_LOSS_BPR) return hsk_bpr_loss_grad(logits, batch, n_cols, loss, grad_logits, ws, stream_);
  HSK_REQUIRE(kind == HSK_LOSS_BCE || kind == HSK_LOSS_SSM, HSK_ERR_INVALID, "unknown loss kind %d", kind);
  HSK_REQUIRE(logits && loss && ws, HSK_ERR_INVALID, "NULL pointer argument");
  HSK_REQUIRE(batch > 0 && n_cols >= 2 && batch < 0x7fffffff, HSK_ERR_INVALID, "need batch > 0 and n_cols >= 2");
  hipStream_t stream = (hipStream_t)stream_;
  const double inv = (kind == HSK_LOSS_BCE) ? 1.0 / ((double)batch * (double)n_cols) : 1.0 / (double)batch;
  k_rec_loss_grad<<<(unsigned)hsk_ceil_div(batch, 4), 256, 0, stream>>>(kind, logits, (int)batch, (long long)n_cols,
                                                                        (float)inv, (float)log_adjust, grad_logits, ws);
  HSK_LAUNCH_CHECK();
  k_loss_mean<<<1, 1024, 0, stream>>>(ws, (int)batch, inv, loss);
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}
