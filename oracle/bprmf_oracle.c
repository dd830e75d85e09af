/*
 * bprmf_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, single thread) of the
 * reference's BPR-MF hot path, used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg as the checker.  Nothing in hassaku_amd/ may link, import or call this file.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks every function below against golden
 * vectors produced by importing the reference itself (oracle/gen_golden.py -> tests/golden/*.npz).
 *
 * Each function cites the reference code it restates (paths relative to the reference tree).
 * Arithmetic: tensors that are fp32 in the reference are fp32 here; reductions are carried in
 * double and rounded once (the reference's fp32 pairwise/vectorised sums are within 1e-6 of that),
 * AdamW is evaluated op by op in fp32 exactly in torch's order (build with -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* algorithms/sgd_alg.py:148-179 -- get_user/item_representations + combine_user_item_representations:
 * out[b,k] = sum_d U[u_b,d]*I[i_bk,d]; out += u_bias; out += i_bias; out += global_bias (that order). */
void orc_mf_scores(const float* U, const float* I, const float* Ib, const float* Ub, const float* gb,
                   int64_t D, const int64_t* u_idx, const int64_t* i_idx, int64_t B, int64_t K, float* logits) {
  for (int64_t b = 0; b < B; ++b) {
    const float* u = U + u_idx[b] * D;
    for (int64_t k = 0; k < K; ++k) {
      const int64_t it = i_idx[b * K + k];
      const float* r = I + it * D;
      double s = 0.0;
      for (int64_t d = 0; d < D; ++d) s += (double)(u[d] * r[d]); /* fp32 product (broadcast mul), then sum */
      float o = (float)s;
      if (Ub) o += Ub[u_idx[b]];
      if (Ib) o += Ib[it];
      if (gb) o += gb[0];
      logits[b * K + k] = o;
    }
  }
}

/* train/rec_losses.py:68-88 -- diff = pos - neg (fp32); BCEWithLogits(diff, 1) with fp64 labels =>
 * fp64 mean over B*N of max(-x,0) + log1p(exp(-|x|)).  grad: d/dx = (sigmoid(x) - 1)/(B*N), routed to
 * logits[:,0] (sum over n) and -that to logits[:,1+n]; cast to fp32 like autograd does. */
double orc_bpr_loss_grad(const float* logits, int64_t B, int64_t K, float* grad /* nullable */) {
  const int64_t N = K - 1;
  const double inv = 1.0 / ((double)B * (double)N);
  double loss = 0.0;
  for (int64_t b = 0; b < B; ++b) {
    const float s0 = logits[b * K];
    double g0 = 0.0;
    for (int64_t n = 0; n < N; ++n) {
      const float xf = s0 - logits[b * K + 1 + n];
      const double x = (double)xf;
      loss += fmax(-x, 0.0) + log1p(exp(-fabs(x)));
      const double gx = (1.0 / (1.0 + exp(-x)) - 1.0) * inv; /* d loss / d x */
      if (grad) grad[b * K + 1 + n] = (float)(-gx);
      g0 += (double)(float)gx;
    }
    if (grad) grad[b * K] = (float)g0;
  }
  return loss * inv;
}

/* train/rec_losses.py:27-53 -- RecBinaryCrossEntropy: BCEWithLogits(logits.flatten(), labels.flatten()) with fp64
 * labels (column 0 = 1): fp64 mean over B*K of softplus(-s) for the positive, softplus(s) for a negative;
 * grad = (sigmoid(s) - y)/(B*K) cast to fp32. */
double orc_bce_loss_grad(const float* logits, int64_t B, int64_t K, float* grad /* nullable */) {
  const double inv = 1.0 / ((double)B * (double)K);
  double loss = 0.0;
  for (int64_t b = 0; b < B; ++b)
    for (int64_t k = 0; k < K; ++k) {
      const double s = (double)logits[b * K + k];
      const double y = (k == 0) ? 1.0 : 0.0;
      loss += fmax(s, 0.0) - s * y + log1p(exp(-fabs(s)));
      if (grad) grad[b * K + k] = (float)((1.0 / (1.0 + exp(-s)) - y) * inv);
    }
  return loss * inv;
}

/* train/rec_losses.py:91-139 -- RecSampledSoftmaxLoss: logits[:,1:] += log_adjust (= log(n_items/neg_train) for
 * uniform sampling); loss = mean_b( -logits[b,0] + logsumexp(logits[b,:]) ), all in fp32 in the reference;
 * grad[b,k] = (softmax_k - [k==0]) / B. */
double orc_ssm_loss_grad(const float* logits, int64_t B, int64_t K, double log_adjust, float* grad /* nullable */) {
  double loss = 0.0;
  const float adj = (float)log_adjust;
  for (int64_t b = 0; b < B; ++b) {
    double mx = -INFINITY;
    for (int64_t k = 0; k < K; ++k) {
      const double z = (double)(k == 0 ? logits[b * K] : logits[b * K + k] + adj);
      if (z > mx) mx = z;
    }
    double sum = 0.0;
    for (int64_t k = 0; k < K; ++k) sum += exp((double)(k == 0 ? logits[b * K] : logits[b * K + k] + adj) - mx);
    loss += -(double)logits[b * K] + mx + log(sum);
    if (grad)
      for (int64_t k = 0; k < K; ++k) {
        const double z = (double)(k == 0 ? logits[b * K] : logits[b * K + k] + adj);
        grad[b * K + k] = (float)((exp(z - mx) / sum - (k == 0 ? 1.0 : 0.0)) / (double)B);
      }
  }
  return loss / (double)B;
}

/* autograd of orc_mf_scores (train/trainer.py:146; embedding_dense_backward scatter-add):
 * dU[u_b] += sum_k g[b,k] I[i_bk]; dI[i_bk] += g[b,k] U[u_b]; dIb[i_bk] += g[b,k];
 * dUb[u_b] += sum_k g[b,k]; dgb += sum g.  Outputs are dense and fully overwritten. */
void orc_mf_backward(const float* U, const float* I, int64_t n_users, int64_t n_items, int64_t D,
                     const int64_t* u_idx, const int64_t* i_idx, int64_t B, int64_t K, const float* g, float* gU,
                     float* gI, float* gIb, float* gUb, float* ggb) {
  /* fp32 accumulation in the reference's order: the per-(b) row gradient (grad_out[:,:,None]*i).sum(1) is
   * summed over k first, then embedding_dense_backward adds rows in flattened (b,k) order.  Adam turns the
   * rounding of near-zero gradient elements into O(lr*1e-5) parameter differences, so the accumulation
   * precision is part of what the fixtures pin. */
  float* aU = (float*)calloc((size_t)(n_users * D), sizeof(float));
  float* aI = (float*)calloc((size_t)(n_items * D), sizeof(float));
  float* aIb = (float*)calloc((size_t)n_items, sizeof(float));
  float* aUb = (float*)calloc((size_t)n_users, sizeof(float));
  float* tb = (float*)malloc((size_t)D * sizeof(float));
  float agb = 0.f;
  for (int64_t b = 0; b < B; ++b) {
    const int64_t u = u_idx[b];
    const float* ur = U + u * D;
    float gsum = 0.f;
    for (int64_t d = 0; d < D; ++d) tb[d] = 0.f;
    for (int64_t k = 0; k < K; ++k) {
      const int64_t it = i_idx[b * K + k];
      const float gv = g[b * K + k];
      const float* ir = I + it * D;
      for (int64_t d = 0; d < D; ++d) {
        tb[d] += gv * ir[d];
        aI[it * D + d] += gv * ur[d];
      }
      aIb[it] += gv;
      gsum += gv;
    }
    for (int64_t d = 0; d < D; ++d) aU[u * D + d] += tb[d];
    aUb[u] += gsum;
    agb += gsum;
  }
  if (gU) memcpy(gU, aU, (size_t)(n_users * D) * sizeof(float));
  if (gI) memcpy(gI, aI, (size_t)(n_items * D) * sizeof(float));
  if (gIb) memcpy(gIb, aIb, (size_t)n_items * sizeof(float));
  if (gUb) memcpy(gUb, aUb, (size_t)n_users * sizeof(float));
  if (ggb) ggb[0] = agb;
  free(aU); free(aI); free(aIb); free(aUb); free(tb);
}

/* torch.optim.AdamW, single-tensor path (train/trainer.py:52-53,147), defaults amsgrad=False:
 *   p.mul_(1 - lr*wd); m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, value=1-b2);
 *   denom = v.sqrt()/sqrt(1-b2^t) + eps; p.addcdiv_(m, denom, value=-(lr/(1-b1^t)))
 * python-float scalars are doubles rounded to fp32 when they meet the fp32 tensor.  g==NULL: zeros. */
void orc_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double b1, double b2,
                    double eps, double wd, int64_t step) {
  const float decay = (float)(1.0 - lr * wd);
  const float w1 = (float)(1.0 - b1);
  const float fb2 = (float)b2;
  const float w2 = (float)(1.0 - b2);
  const float step_size = (float)(lr / (1.0 - pow(b1, (double)step)));
  const float bc2s = (float)sqrt(1.0 - pow(b2, (double)step));
  const float feps = (float)eps;
  for (int64_t i = 0; i < n; ++i) {
    const float gi = g ? g[i] : 0.f;
    float pi = p[i] * decay;
    float mi = m[i] + w1 * (gi - m[i]);
    float vi = v[i] * fb2;
    vi = vi + (w2 * gi) * gi;
    const float denom = sqrtf(vi) / bc2s + feps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

/* torch.optim.Adam single-tensor step as train/trainer.py:48-49 configures it (weight_decay = L2):
 *   g = g.add(p, alpha=wd); m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, value=1-b2);
 *   denom = v.sqrt()/sqrt(1-b2^t) + eps; p.addcdiv_(m, denom, value=-(lr/(1-b1^t)))          (no decoupled decay) */
void orc_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double b1, double b2,
                   double eps, double wd, int64_t step) {
  const float fwd = (float)wd;
  const float w1 = (float)(1.0 - b1);
  const float fb2 = (float)b2;
  const float w2 = (float)(1.0 - b2);
  const float step_size = (float)(lr / (1.0 - pow(b1, (double)step)));
  const float bc2s = (float)sqrt(1.0 - pow(b2, (double)step));
  const float feps = (float)eps;
  for (int64_t i = 0; i < n; ++i) {
    float gi = g ? g[i] : 0.f;
    gi = fmaf(fwd, p[i], gi);   /* ATen's add(alpha) kernel is vec::fmadd(b, alpha, a) */
    float mi = m[i] + w1 * (gi - m[i]);
    float vi = v[i] * fb2;
    vi = vi + (w2 * gi) * gi;
    const float denom = sqrtf(vi) / bc2s + feps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi; v[i] = vi;
  }
}

/* torch.optim.Adagrad single-tensor step as train/trainer.py:50-51 configures it (lr_decay 0, accumulator 0, eps 1e-10):
 *   g = g.add(p, alpha=wd); clr = lr; state_sum.addcmul_(g, g, value=1); std = state_sum.sqrt().add_(eps);
 *   p.addcdiv_(g, std, value=-clr) */
void orc_adagrad_step(float* p, const float* g, float* sum, int64_t n, double lr, double eps, double wd) {
  const float fwd = (float)wd, clr = (float)lr, feps = (float)eps;
  for (int64_t i = 0; i < n; ++i) {
    float gi = g ? g[i] : 0.f;
    gi = fmaf(fwd, p[i], gi);   /* ATen's add(alpha) kernel is vec::fmadd(b, alpha, a) */
    const float si = sum[i] + gi * gi;
    const float std = sqrtf(si) + feps;
    p[i] = p[i] - clr * (gi / std);
    sum[i] = si;
  }
}

/* data/dataloader.py:114-124 invariant checker: every negative is in [0,n_items) and not in the
 * user's CSR row.  Returns the number of violations. */
int64_t orc_count_bad_negatives(const int64_t* indptr, const int32_t* indices, int64_t n_items,
                                const int64_t* u_idx, const int64_t* neg, int64_t B, int64_t N) {
  int64_t bad = 0;
  for (int64_t b = 0; b < B; ++b) {
    const int64_t lo = indptr[u_idx[b]], hi = indptr[u_idx[b] + 1];
    for (int64_t n = 0; n < N; ++n) {
      const int64_t x = neg[b * N + n];
      if (x < 0 || x >= n_items) { ++bad; continue; }
      for (int64_t e = lo; e < hi; ++e)
        if (indices[e] == x) { ++bad; break; }
    }
  }
  return bad;
}

/* eval/eval.py:240-251 -- scores of R users against ALL items + biases, then -inf on the exclude CSR. */
void orc_eval_scores(const float* U, const float* I, const float* Ib, const float* Ub, const float* gb,
                     int64_t n_items, int64_t D, const int64_t* u_idx, int64_t R, const int64_t* excl_indptr,
                     const int32_t* excl_indices, float* out) {
  for (int64_t r = 0; r < R; ++r) {
    const int64_t u = u_idx[r];
    const float* ur = U + u * D;
    for (int64_t it = 0; it < n_items; ++it) {
      const float* ir = I + it * D;
      double s = 0.0;
      for (int64_t d = 0; d < D; ++d) s += (double)(ur[d] * ir[d]);
      float o = (float)s;
      if (Ub) o += Ub[u];
      if (Ib) o += Ib[it];
      if (gb) o += gb[0];
      out[r * n_items + it] = o;
    }
    if (excl_indptr)
      for (int64_t e = excl_indptr[u]; e < excl_indptr[u + 1]; ++e) out[r * n_items + excl_indices[e]] = -INFINITY;
  }
}

/* logits.topk(k) (eval/eval.py:63): k largest per row, descending; ties: lower index first. */
void orc_topk(const float* x, int64_t R, int64_t C, int64_t k, float* vals, int64_t* idx) {
  char* used = (char*)malloc((size_t)C);
  for (int64_t r = 0; r < R; ++r) {
    const float* row = x + r * C;
    memset(used, 0, (size_t)C);
    for (int64_t j = 0; j < k; ++j) {
      int64_t best = -1;
      for (int64_t c = 0; c < C; ++c) {
        if (used[c]) continue;
        if (best < 0 || row[c] > row[best]) best = c;
      }
      used[best] = 1;
      vals[r * k + j] = row[best];
      idx[r * k + j] = best;
    }
  }
  free(used);
}

/* eval/metrics.py:4-105 from a ranked id list and the ground-truth CSR:
 * out[r, t, 0..2] = precision@k_t, recall@k_t (0 if no ground truth), ndcg@k_t (0 if none, clamp<=1),
 * discount 1/log2(rank+2) in fp32 as in the reference's discount_template. */
void orc_rank_metrics(const int64_t* topk, int64_t R, int64_t kmax, const int64_t* u_idx, const int64_t* indptr,
                      const int32_t* indices, const int32_t* ks, int32_t n_ks, float* out) {
  for (int64_t r = 0; r < R; ++r) {
    const int64_t lo = indptr[u_idx[r]], hi = indptr[u_idx[r] + 1];
    const int64_t n_rel = hi - lo;
    for (int32_t t = 0; t < n_ks; ++t) {
      const int k = ks[t];
      float hits = 0.f, dcg = 0.f, idcg = 0.f;
      for (int j = 0; j < k; ++j) {
        const float disc = 1.f / log2f((float)(j + 2));
        const int64_t it = topk[r * kmax + j];
        int hit = 0;
        for (int64_t e = lo; e < hi; ++e)
          if (indices[e] == it) { hit = 1; break; }
        if (hit) { hits += 1.f; dcg += disc; }
        if (j < n_rel) idcg += disc;
      }
      float* o = out + (r * n_ks + t) * 3;
      o[0] = hits / (float)k;
      o[1] = n_rel > 0 ? hits / (float)n_rel : 0.f;
      float nd = n_rel > 0 ? dcg / idcg : 0.f;
      o[2] = nd > 1.f ? 1.f : nd;
    }
  }
}
