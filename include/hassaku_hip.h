/*
 * hassaku_hip.h -- C ABI of libhassaku_hip.so: the MI355X (gfx950) BPR-MF training / full-eval path.
 *
 * The reference (karapostK/hassaku) has no FFI of its own: the seam is a Python plugin API
 * (SURVEY.md section 8b).  Every entry point below names the reference interface it replaces
 * (paths relative to the reference tree).  All pointers are DEVICE pointers unless stated, all
 * tables are dense row-major fp32 with leading dimension == dim (the nn.Embedding layout, so
 * state_dict()/model.pth round-trip unchanged), every call is asynchronous on `stream`
 * (a hipStream_t passed as void*), allocates nothing and returns 0 on success.  On failure a
 * non-zero code is returned and hsk_last_error() holds the text (thread-local).
 *
 * Index arrays on the drop-in surface are int64 (the dtype the reference's loader yields,
 * data/dataloader.py:126-129).  Out-of-range indices never touch memory: they are clamped to row
 * 0 and bit 0 of *status is set (the reference raises IndexError / device assert instead).
 */
#ifndef HASSAKU_HIP_H
#define HASSAKU_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hsk_stream_t; /* hipStream_t */

enum {
  HSK_OK = 0,
  HSK_ERR_INVALID = 1,     /* bad argument (null pointer, negative size, ...) */
  HSK_ERR_UNSUPPORTED = 2, /* shape outside what the kernels are instantiated for */
  HSK_ERR_HIP = 3          /* a HIP runtime call failed */
};

/* loss kinds (hsk_bprmf_state.loss_kind, hsk_rec_loss_grad) */
enum { HSK_LOSS_BPR = 0, HSK_LOSS_BCE = 1, HSK_LOSS_SSM = 2 };

/* optimiser kinds (conf['optimizer'], train/trainer.py:48-53) */
enum {
  HSK_OPT_ADAMW = 0,
  HSK_OPT_ADAM = 1,
  HSK_OPT_ADAGRAD = 2
};

/* bits of the device-side status word */
enum {
  HSK_STATUS_BAD_INDEX = 1,       /* an index was outside [0, n) and was clamped */
  HSK_STATUS_SAMPLER_GAVE_UP = 2  /* rejection sampler hit its retry cap (user owns ~all items) */
};

/* library version (major*10000 + minor*100 + patch) and last error text of the calling thread */
int hsk_version(void);
const char* hsk_last_error(void);
/* number of CUs / wavefront size / gcn arch name of the current device (host pointers) */
int hsk_device_info(int32_t* cu_count, int32_t* wave_size, char* arch, int32_t arch_len);

/* ---------------------------------------------------------------------------------------------
 * Un-fused operators (drop-in for the autograd path of an unmodified Trainer.fit)
 * ------------------------------------------------------------------------------------------ */

/*
 * logits[b,k] = <user_emb[u[b]], item_emb[i[b,k]]> (+ user_bias[u[b]]) (+ item_bias[i[b,k]]) (+ global_bias)
 * Replaces SGDBasedRecommenderAlgorithm.forward (algorithms/base_classes.py:99-108) =
 * SGDMatrixFactorization.get_user_representations / get_item_representations /
 * combine_user_item_representations (algorithms/sgd_alg.py:148-179).
 * Bias pointers may be NULL (bias disabled).  u_idx: [batch], i_idx: [batch, n_cols].
 * dim == 0 (user_emb = item_emb = NULL) is the bias-only model SGDBaseline (algorithms/sgd_alg.py:72-107):
 * logits = user_bias[u] + item_bias[i] + global_bias; hsk_mf_backward accepts the same form.
 */
int hsk_mf_scores(const float* user_emb, const float* item_emb, const float* item_bias,
                  const float* user_bias, const float* global_bias,
                  int64_t n_users, int64_t n_items, int64_t dim,
                  const int64_t* u_idx, const int64_t* i_idx, int64_t batch, int64_t n_cols,
                  float* logits, int32_t* status, hsk_stream_t stream);

/*
 * BPR loss on logits [batch, n_cols] (column 0 = positive):
 *   loss = mean_{b,n} softplus(-(logits[b,0] - logits[b,1+n]))   (fp64 accumulate, fp64 result)
 * and, if grad_logits != NULL, d loss / d logits (fp32, already divided by batch*(n_cols-1)).
 * Replaces RecBayesianPersonalizedRankingLoss.compute_loss (train/rec_losses.py:68-88) and its
 * autograd backward.  ws: device scratch of `batch` doubles.
 */
int hsk_bpr_loss_grad(const float* logits, int64_t batch, int64_t n_cols,
                      double* loss, float* grad_logits, double* ws, hsk_stream_t stream);

/*
 * The same for any of the reference's three losses (train/rec_losses.py:27-53 bce, :56-88 bpr, :91-139
 * sampled_softmax): kind = HSK_LOSS_*; log_adjust is added to the negatives' logits before the softmax
 * (log(n_items / neg_train) under uniform sampling, 0 otherwise; ignored by bpr / bce).
 *   bce: mean over batch*n_cols of BCEWithLogits(logit, [col == 0]);  ssm: mean over batch of -l_0 + logsumexp.
 */
int hsk_rec_loss_grad(int32_t kind, const float* logits, int64_t batch, int64_t n_cols, double log_adjust,
                      double* loss, float* grad_logits, double* ws, hsk_stream_t stream);

/*
 * Dense parameter gradients of hsk_mf_scores given grad_logits [batch, n_cols]
 * (what autograd's embedding_dense_backward produces for train/trainer.py:146).
 * Output buffers are fully overwritten (zero for rows not in the batch).  NULL outputs are skipped.
 */
int hsk_mf_backward(const float* user_emb, const float* item_emb,
                    int64_t n_users, int64_t n_items, int64_t dim,
                    const int64_t* u_idx, const int64_t* i_idx, int64_t batch, int64_t n_cols,
                    const float* grad_logits,
                    float* g_user_emb, float* g_item_emb, float* g_item_bias,
                    float* g_user_bias, float* g_global_bias,
                    int32_t* status, hsk_stream_t stream);

/*
 * Embedding gather and its dense backward for the reference's other SGD models (nn.Embedding in ACF / UProtoMF /
 * IProtoMF / UIProtoMF, algorithms/sgd_alg.py:187-570): out[j] = table[idx[j]];  grad_table[r] = sum of grad_out[j]
 * over the positions with idx[j] == r, added in ascending j (deterministic, no atomics), zero for rows nobody named --
 * what embedding_dense_backward returns for nn.Embedding(sparse=False).  ws: hsk_embedding_backward_ws_bytes(n_rows, n).
 */
int hsk_embedding_gather(const float* table, int64_t n_rows, int64_t dim, const int64_t* idx, int64_t n,
                         float* out, int32_t* status, hsk_stream_t stream);
int64_t hsk_embedding_backward_ws_bytes(int64_t n_rows, int64_t n);
int hsk_embedding_backward(const float* grad_out, const int64_t* idx, int64_t n, int64_t n_rows, int64_t dim,
                           float* grad_table, void* ws, int64_t ws_bytes, int32_t* status, hsk_stream_t stream);

/*
 * One dense AdamW step on a flat parameter of n elements -- torch.optim.AdamW defaults
 * (train/trainer.py:52-53,147): p*=1-lr*wd; m=lerp(m,g,1-b1); v=b2*v+(1-b2)g^2;
 * p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  `step` is t (1-based, after increment).
 * g == NULL means an all-zero gradient.
 */
int hsk_adamw_dense(float* p, const float* g, float* m, float* v, int64_t n,
                    double lr, double beta1, double beta2, double eps, double wd, int64_t step,
                    hsk_stream_t stream);

/*
 * The same for any of the three optimisers train/trainer.py:48-53 can select (torch defaults otherwise):
 *   HSK_OPT_ADAMW    as above
 *   HSK_OPT_ADAM     torch.optim.Adam, weight_decay as L2: g += wd*p, then the Adam update without decay
 *   HSK_OPT_ADAGRAD  torch.optim.Adagrad (lr_decay 0, initial accumulator 0): g += wd*p; v += g*g;
 *                    p -= lr * g / (sqrt(v) + eps); v = state_sum; m keeps its values but must be valid memory
 */
int hsk_opt_dense(int opt_kind, float* p, const float* g, float* m, float* v, int64_t n,
                  double lr, double beta1, double beta2, double eps, double wd, int64_t step,
                  hsk_stream_t stream);

/*
 * Uniform negative sampling with rejection of the user's training positives.
 * Replaces NegativeSampler._neg_sample_uniform + TrainDataLoader._neg_sampling_collate_fn
 * (data/dataloader.py:56-57,92-129): neg[b,n] ~ U{[0,n_items) \ csr_row(u[b])}, i.i.d. with
 * replacement.  csr_indices must be sorted inside each row.  RNG: Philox4x32-10 keyed by `seed`,
 * counter (stream_id, b, n, attempt) -- reproducible and independent of launch geometry.
 */
int hsk_sample_negatives_uniform(const int64_t* csr_indptr, const int32_t* csr_indices,
                                 int64_t n_users, int64_t n_items,
                                 const int64_t* u_idx, int64_t batch, int64_t n_neg,
                                 uint64_t seed, uint64_t stream_id,
                                 int64_t* neg_out, int32_t* status, hsk_stream_t stream);

/*
 * The same with negatives drawn from an item distribution p (train_neg_strategy 'popular':
 * NegativeSampler._neg_sample_popular, data/dataloader.py:59-64, p = pop_distribution^squash normalised), given
 * as a Walker alias table built on the host: column j keeps itself with probability alias_prob[j], else
 * alias_idx[j].  Still rejected against the user's CSR row, i.e. p restricted to the items the user has not seen.
 */
int hsk_sample_negatives_alias(const int64_t* csr_indptr, const int32_t* csr_indices,
                               int64_t n_users, int64_t n_items,
                               const float* alias_prob, const int32_t* alias_idx,
                               const int64_t* u_idx, int64_t batch, int64_t n_neg,
                               uint64_t seed, uint64_t stream_id,
                               int64_t* neg_out, int32_t* status, hsk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused BPR-MF AdamW training step (the hot path of Trainer.fit, train/trainer.py:128-148)
 * ------------------------------------------------------------------------------------------ */

typedef struct hsk_bprmf_state {
  /* ---- caller-owned: parameters, optimiser state, shapes ------------------------------------------------------- */
  /* parameters: user_emb [U,D], item_emb [I,D], item_bias [I], user_bias [U], global_bias [1];
     bias pointers NULL when disabled (algorithms/sgd_alg.py:127-138) */
  float* user_emb;
  float* item_emb;
  float* item_bias;
  float* user_bias;
  float* global_bias;
  /* AdamW first/second moments, same shapes (torch.optim.AdamW state 'exp_avg','exp_avg_sq') */
  float* m_user_emb;   float* v_user_emb;
  float* m_item_emb;   float* v_item_emb;
  float* m_item_bias;  float* v_item_bias;
  float* m_user_bias;  float* v_user_bias;
  float* m_global_bias; float* v_global_bias;
  int64_t n_users, n_items, dim;
  /* hyper-parameters (conf keys lr, wd; betas/eps = torch defaults unless overridden).  FROZEN once
     hsk_bprmf_init_workspace has run (the per-step Adam scalars of the lazy replay and of replayed graphs are a device
     table computed from them): a call that finds lr / betas / eps / wd / opt_kind changed fails with HSK_ERR_INVALID
     instead of mixing two schedules -- flush, set the new values, call hsk_bprmf_init_workspace again */
  double lr, beta1, beta2, eps, wd;
  /* number of optimizer steps applied so far; hsk_bprmf_train_step* increments it (host side) */
  int64_t step;
  /* training interactions: CSR by user (sorted rows) for the sampler; COO for iteration */
  const int64_t* csr_indptr;   /* [U+1] */
  const int32_t* csr_indices;  /* [nnz] */
  const int32_t* coo_user;     /* [nnz] */
  const int32_t* coo_item;     /* [nnz] */
  int64_t nnz;
  uint64_t seed;
  /* scratch, sized by hsk_bprmf_workspace_bytes, prepared once by hsk_bprmf_init_workspace */
  void* workspace;
  int64_t workspace_bytes;
  int64_t max_batch, max_cols;
  /* ---- caller-owned: options ----------------------------------------------------------------------------------- */
  /* lazy user-table AdamW: 0 = dense sweep every step (reference order of operations),
     1 = exact lazy catch-up of untouched rows (needs hsk_bprmf_flush before reading user tables) */
  int32_t lazy_users;
  /* bit s set: bracket stage s of the step with HIP events (see HSK_STAGE_*, hsk_timing_*) */
  int32_t timing_mask;
  /* opaque handle from hsk_timing_create, or NULL (no timing) */
  void* timing;
  /* opaque handle from hsk_aux_create, or NULL: side stream on which the batch named by hsk_bprmf_hint_next is
     sampled and item-sorted while the current step's item / user passes run (fork / join by events) */
  void* aux;
  /* event-time only every timing_every-th step (<= 1: every step) */
  int32_t timing_every;
  /* recommendation loss of the fused step (train/rec_losses.py): HSK_LOSS_BPR (default, 0), HSK_LOSS_BCE,
     HSK_LOSS_SSM (sampled softmax; ssm_log_adjust = log(n_items / neg_train) for uniform sampling, else 0) */
  int32_t loss_kind;
  /* optimiser selected by conf['optimizer'] (train/trainer.py:48-53): HSK_OPT_ADAMW (default, 0), HSK_OPT_ADAM,
     HSK_OPT_ADAGRAD.  m_* = exp_avg (unused by adagrad, must still be valid memory), v_* = exp_avg_sq /
     adagrad's state_sum.  torch defaults: eps 1e-8 (adam, adamw), 1e-10 (adagrad); betas only for adam / adamw */
  int32_t opt_kind;
  /* 1: lazy, exact AdamW on the ITEM tables too (same scheme as lazy_users: a row outside the batch keeps its
     zero-gradient steps until it is next touched or flushed; bit-identical to the dense update).  For catalogues far
     larger than a batch touches; needs an even dim.  0 (default): every item row is updated every step */
  int32_t lazy_items;
  double ssm_log_adjust;
  /* negative sampling law of the device sampler: NULL = uniform; otherwise a Walker alias table over the items
     (alias_prob float[I], alias_idx int32[I]) = train_neg_strategy 'popular' (data/dataloader.py:59-64) */
  const float* alias_prob;
  const int32_t* alias_idx;
  /* hsk_bprmf_train_steps replays its steady-state loop as captured HIP graphs of this many steps (0: default 64,
     < 0: never, eager launches only); per-step scalars are read from a device descriptor, results are bit-identical */
  int32_t graph_chunk;
  /* lazy_users: where the pending zero-gradient AdamW steps of the batch's user rows are replayed.  0 (default): inside
     the forward kernel, in registers (fastest step); 1: by a stand-alone launch in front of it (the forward is then a
     pure gather -- what bench.py times as `roofline.pure_gather`); results are bit-identical */
  int32_t catchup_apart;
  /* outputs: loss_out[0] = loss of the last step (fp64), loss_out[1] += that loss (epoch sum) */
  double* loss_out;
  int32_t* status;
  /* lazy AdamW: every flush_every-th step sweeps the lazily updated tables (every row brought up to the current step),
     which bounds how many zero-gradient steps a row replays when it is next touched.  A speed matter only -- any
     cadence is exact.  0: chosen from the table and batch sizes (hsk_bprmf_flush_cadence) */
  int32_t flush_every;
  /* 1: workspace carved for the sharded step (hsk_shard_*), which keeps the batch's user rows and their gradients in
     its exchange buffers: the [max_batch, dim] row buffers of the single-GPU step are not allocated.  The size is then
     hsk_shard_base_workspace_bytes(); the single-GPU entry points refuse such a state */
  int32_t ws_sharded;
  /* ---- LIBRARY SCRATCH: zero-initialise, never write ----------------------------------------------------------- */
  /* written by hsk_bprmf_init_workspace: the hyper-parameters (and optimiser) the workspace was prepared for */
  double frozen_hyper[5];
  int32_t frozen_opt;
  int32_t frozen_valid;
  /* whether the step being issued is an event-timed one (from timing_every) */
  int32_t timing_now;
  int32_t reserved3;
} hsk_bprmf_state;

int64_t hsk_bprmf_workspace_bytes(int64_t n_users, int64_t n_items, int64_t dim,
                                  int64_t max_batch, int64_t max_cols);
/* Steps between two sweeps of a lazily updated table (table 0: users, 1: items) of which a step touches about
 * `touched_rows` rows, as the step picks it when st->flush_every == 0.  A sweep moves 24 bytes per table element; a
 * touched row replays its pending zero-gradient steps (VALU work, ~3.3e-13 s per element and step).  The cadence that
 * minimises the sum: ~sqrt(rows / touched_rows) * 5 while a row waits much longer than that for its next batch (a
 * cfg5 shard: ~210), never (2^30: only an explicit flush sweeps) when a row is touched every few steps anyway
 * (ml10m: a user is in every 17th batch). */
int32_t hsk_bprmf_flush_cadence(const hsk_bprmf_state* st, int32_t table, int64_t touched_rows);
/* prepares the workspace for st's shapes and hyper-parameters (synchronous); st->step rows are taken as current */
int hsk_bprmf_init_workspace(hsk_bprmf_state* st, hsk_stream_t stream);

/* One step on a loader-provided batch (u_idx [batch], i_idx [batch, n_cols], column 0 positive):
 * forward + BPR loss + backward + AdamW on every parameter, results equal to the reference's
 * out=model(u,i); loss=bpr(out,labels); loss.backward(); optimizer.step() on the same batch. */
int hsk_bprmf_train_step(hsk_bprmf_state* st, const int64_t* u_idx, const int64_t* i_idx,
                         int64_t batch, int64_t n_cols, hsk_stream_t stream);

/* Same step with the batch built on the device: positives = interactions order[start .. start+batch)
 * of the COO (order == NULL: identity), n_neg negatives each from the on-device rejection sampler. */
int hsk_bprmf_train_step_sampled(hsk_bprmf_state* st, const int64_t* order, int64_t start,
                                 int64_t batch, int64_t n_neg, hsk_stream_t stream);

/* Stages of one fused step, in launch order (timing slots). */
enum {
  HSK_STAGE_PREP = 0,      /* batch -> int32 copies / device sampler, histogram, owner map */
  HSK_STAGE_SCAN = 1,      /* exclusive scan of the item histogram */
  HSK_STAGE_SCATTER = 2,   /* item-major permutation */
  HSK_STAGE_FWD = 3,       /* gather + scores + BPR + user-row gradient (the roofline kernel) */
  HSK_STAGE_ITEM = 4,      /* item-major gradient reduction + AdamW on item rows */
  HSK_STAGE_USER = 5,      /* user-table AdamW */
  HSK_STAGE_FINISH = 6,    /* loss reduction, global bias */
  HSK_STAGE_COUNT = 7
};

/* Per-stage device timing with HIP events recorded on the stream the kernels are launched on.
 * hsk_timing_create/destroy/collect are host-synchronous helpers and must not be called while the
 * stream is being captured.  collect() waits for the recorded events, adds the elapsed
 * milliseconds and the number of samples of every stage into ms_sum[HSK_STAGE_COUNT] /
 * count[HSK_STAGE_COUNT] (host arrays), and resets the recorder. */
void* hsk_timing_create(void);
void hsk_timing_destroy(void* timing);
int hsk_timing_collect(void* timing, double* ms_sum, int64_t* count);

/* Side stream + events for cross-step overlap (host-side objects; create once per state, destroy after the
 * last step has completed).  Returns NULL on failure. */
void* hsk_aux_create(void);
void hsk_aux_destroy(void* aux);

/* Name the batch the NEXT hsk_bprmf_train_step_sampled call will ask for (same order pointer, start, batch,
 * n_neg).  The step issued after the hint samples and sorts that batch on the aux side stream, behind its own
 * forward kernel, into a second set of workspace buffers; the next call finds it ready.  Purely a scheduling hint:
 * the batch, its negatives (RNG stream id = the step index it is consumed at) and every result are identical to
 * the un-hinted sequence, and a hint that turns out wrong is discarded.  The data loader's epoch loop
 * (data/dataloader.py:92-129 feeding train/trainer.py:127-160 in the reference) knows its next batch, which is what
 * makes this the device-side counterpart of DataLoader prefetching.  batch <= 0 clears a pending hint.
 * Needs st->aux; `order` must stay valid and unchanged until the hinted step has been issued. */
int hsk_bprmf_hint_next(hsk_bprmf_state* st, const int64_t* order, int64_t start, int64_t batch, int64_t n_neg);

/* n_steps consecutive hsk_bprmf_train_step_sampled calls on the batches order[start + s*batch .. +batch), s < n_steps,
 * each hinting the next one to the prefetch (when st->aux is set): the inner loop of an epoch (train/trainer.py:128-148)
 * issued from C.  At small batches the step is bound by the host's launch rate (ten HIP calls of 3-4 us each plus the
 * interpreter), not by the GPU: with st->aux set (and an even dim, no lazy_items, no stage timing) runs of consecutive
 * steps are captured once as a HIP graph and REPLAYED (st->graph_chunk steps per graph), the per-step scalars coming
 * from a device-resident descriptor; same kernels, same order, bit-identical results.
 * A captured graph has the state frozen into its kernel arguments (every pointer, the shapes, lr / wd / betas / eps,
 * seed, loss and optimiser kind, the lazy flags); it is replayed only while the state is byte-for-byte what it was at
 * capture time, `step` and the timing fields aside -- change anything else between two calls (an LR schedule through
 * st->lr, rebound parameters, another dataset) and the run is captured afresh, as the eager path would see it. */
int hsk_bprmf_train_steps(hsk_bprmf_state* st, const int64_t* order, int64_t start, int64_t n_steps, int64_t batch,
                          int64_t n_neg, hsk_stream_t stream);

/* Name the batch that FOLLOWS the next hsk_bprmf_train_steps run (the first batch of the caller's next run): the run's
 * last step then prepares it on the side stream, as every other step of the run does for its successor -- an epoch
 * loop issued in several runs (data/dataloader.py:92-129 feeding train/trainer.py:127-160) keeps its prefetch pipeline
 * full across the calls.  Same rules as hsk_bprmf_hint_next; consumed by (and cleared after) the next run; ignored by
 * runs replayed as graphs, which prepare their first batch themselves.  batch <= 0 clears it. */
int hsk_bprmf_hint_after_run(hsk_bprmf_state* st, const int64_t* order, int64_t start, int64_t batch, int64_t n_neg);
/* The same, naming the n_batches consecutive batches order[start + j*batch .. +batch), j < n_batches, that follow the
 * next run.  Large batches (the item-partitioned forward) prepare a batch over the TWO steps in front of its own -- its
 * sampling and the phases of its item sort ride as extra workgroups in those steps' launches (csrc/hsk_fused.hip:
 * hsk_pipe_step) -- so an epoch loop issued in several runs names two batches to keep that pipeline full across the
 * calls; hsk_bprmf_hint_after_run names one. */
int hsk_bprmf_hint_after_run_n(hsk_bprmf_state* st, const int64_t* order, int64_t start, int64_t batch, int64_t n_neg,
                               int64_t n_batches);

/* number of runs hsk_bprmf_train_steps has issued as replayed graphs so far (0 when every step was launched eagerly) */
int64_t hsk_bprmf_graph_replays(const hsk_bprmf_state* st);
/* number of steps hsk_bprmf_train_steps has issued with the next batches' preparation riding in the steps' own launches
 * (large batches; 0 when every step used the side-stream prefetch) */
int64_t hsk_bprmf_pipelined_steps(const hsk_bprmf_state* st);
/* Process-wide switch of that in-launch preparation (1: on, the default; environment HSK_PIPE sets the initial value):
 * off, every step prepares its next batch on the side stream.  Results are bit-identical either way. */
void hsk_bprmf_set_pipeline(int on);

/* Bring lazily-updated user / item rows up to st->step (no-op with dense updates); also drops a pending hint and a
 * prefetched batch that was never trained on (a flush ends a run of steps). */
int hsk_bprmf_flush(hsk_bprmf_state* st, hsk_stream_t stream);

/* Copy of the device batch the last *_sampled step used (debug / parity): u [batch], items [batch,n_cols] */
int hsk_bprmf_last_batch(const hsk_bprmf_state* st, int64_t batch, int64_t n_cols,
                         int64_t* u_out, int64_t* i_out, hsk_stream_t stream);

/* Arithmetic of the score GEMMs of hsk_mf_eval_topk[_planes] and hsk_mf_eval_topk_fused:
 *   2 (default)  every fp32 operand scaled by a power of two (its table's largest |x| -> [2^14, 2^15)) and cut into two fp16
 *                pieces, three fp16 MFMAs per product block on the 256 x 256 kernels -- error against float64 at or below
 *                that of an fp32 GEMM (<= 1e-6 of the largest score; pieces exact for every element within 2^-18 of the
 *                table's maximum), at half the MFMAs of form 1.  Taken wherever the call brings the pieces' scratch
 *                (hsk_mf_eval_planes_bytes / hsk_mf_eval_fused_ws_bytes_dim) and dim % 4 == 0 with 16-byte aligned
 *                tables; any other call runs form 1;
 *   1            three bf16 pieces per operand, six bf16 MFMAs per product block (no scaling: the whole fp32 range);
 *   0            the exact-fp32 MFMA form.
 * Both entry points follow the switch and the same order of operations, so for one set of arguments (scratch given to both
 * or to neither) the materialised and the fused path agree bit for bit in every form.
 * Process-wide; the environment variable HSK_EVAL_X3 sets the initial value.  (Replaces nothing in the reference: its
 * scores are torch's fp32 matmul, eval/eval.py:240-248; all three forms meet the 1e-5 bound against it.) */
void hsk_eval_set_arith(int form);

/* Columns of the step's internal batch rows for a batch of this shape: n_cols, or n_cols + P - 1 when the step runs the
 * item-partitioned forward (csrc/hsk_fwd_part.h: large batches, item tables of a few L2 sizes), whose rows hold the
 * positive item P times -- unit q of the forward leaves its share of d loss/d s_0 in column q.  The entry index e of
 * hsk_bprmf_last_sort counts in these columns.  (The reference has no counterpart: its batch is the loader's
 * [B, 1+N] tensor, data/dataloader.py:92-129.) */
int64_t hsk_bprmf_batch_columns(const hsk_bprmf_state* st, int64_t batch, int64_t n_cols);

/* Copy of the item-major index the last step built from its batch (debug / parity): perm int32[n_entries] = the
 * batch entries e = b*cols + k (cols = hsk_bprmf_batch_columns) grouped by item, ascending e inside an item; offsets
 * int32[n_items + 1] = where each item's entries start in perm.  Three kernel chains build it depending on the shape
 * (csrc/hsk_sort.h); all must return exactly this. */
int hsk_bprmf_last_sort(const hsk_bprmf_state* st, int64_t n_entries, int32_t* perm_out, int32_t* offsets_out,
                        hsk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU fused step: one process per GPU; replaces nn.DataParallel (train/trainer.py:38-41).
 *
 * Both tables are SHARDED, nothing is replicated and no dense gradient is ever reduced:
 *   items  range-sharded: rank r owns items [item_lo, item_lo + I_loc) with their AdamW moments
 *          (base.item_emb / item_bias hold only those rows, base.n_items = I_loc);
 *   users  row-sharded:   rank r owns the users u with u % world == r at local row u / world
 *          (base.user_emb holds only those rows, base.n_users = number of local rows).
 * A step processes a GLOBAL batch of G = world * batch positives.  Every rank draws the same Philox stream of
 * negatives for the whole global batch and KEEPS THE ENTRIES WHOSE ITEM IT OWNS (exactly the single-GPU samples;
 * no item row ever moves).  What moves are the batch's user rows and their gradients:
 *
 *   hsk_shard_prepare      batch ids only (may run a step ahead on a side stream, buffer set `set`): sample + keep
 *                          the owned entries (compact, per positive in column order), slot of every positive's user
 *                          at its owner (slot_of_b = owner*C + position in batch order; every rank computes the whole
 *                          map, so no request exchange), item sort of the kept entries, owner map of the requested rows
 *   hsk_shard_pack         owner: replay the missed zero-gradient AdamW steps of the requested user rows, pack them
 *     all_gather(rows_send [C,D] -> rows_all [world*C, D])                     4*D*C bytes to every peer
 *   hsk_shard_pos_scores   s0[b] = <u_b, i_b0> + bias for the positives whose item this rank owns (0 elsewhere);
 *                          lazily updated item rows of the batch are brought up to date first
 *     all_reduce(s0 [G], sum)                                                   4*G bytes
 *   hsk_shard_forward      per positive: scores of the OWNED negatives, d loss/d score, partial user-row gradient
 *                          (-> dU_all[slot]), partial sum of the negatives' weights (-> gsum[b]), partial loss
 *     all_reduce(gsum [G], sum)                                                 4*G bytes
 *   hsk_shard_pos_fix      owner of the positive item: g_0 = -gsum[b]; dU_all[slot] += g_0 * i_b0
 *     reduce_scatter(dU_all [world*C, D] -> grads_mine [C, D], sum)            4*D*C bytes to every peer
 *   hsk_shard_apply_items  item-major gradient reduction + AdamW on the local item shard -- overlaps the reduce_scatter
 *   hsk_shard_apply_users  owner: AdamW on the requested user rows (duplicates summed in slot order) [needs grads_mine];
 *                          loss_out[0] = this rank's share of the global mean loss (sum over ranks = the loss),
 *                          loss_out[1] accumulates it.  Closes the step.
 *
 * Result = the single-GPU step on the global batch (same samples; the fp32 sums over a positive's negatives are split
 * by owner, so the order of summation differs).  world == 1 is accepted (the collectives degenerate to copies): the
 * same code path on one GPU.
 *
 * Losses (base.loss_kind; nn.DataParallel in the reference is loss-agnostic, train/trainer.py:38-41 with
 * train/rec_losses.py:27-139).  The protocol above is the BPR one.  Differences:
 *   HSK_LOSS_BCE  every logit is weighed on its own: NO scalar collective at all -- the positive's owner uses its own s0
 *                 (skip both all_reduces; hsk_shard_pos_fix adds the positive's term from sh->s0 as it stands)
 *   HSK_LOSS_SSM  sampled softmax: hsk_shard_pos_scores leaves the owner's s0 in the third plane of ssm_send,
 *                 hsk_shard_forward the rank's running max and normaliser over ITS negatives in the first two;
 *                   all_gather(ssm_send [3*G] -> ssm_all [world, 3*G])           12*G bytes to every peer
 *                 replaces both all_reduces; hsk_shard_pos_fix then combines the pairs (identically on every rank),
 *                 scales the rank's partial gradient rows and weights, and adds the positive's term at its owner.
 *
 * Capacities.  C = user slots per owner, entry_cap = kept entries per rank; if either is too small for a batch
 * HSK_STATUS_SHARD_OVERFLOW is raised in *status when that batch is PREPARED, and when its step opens (hsk_shard_pack)
 * every table-writing kernel of that step AND OF EVERY LATER STEP returns without writing (a device-side guard, one scalar
 * load per workgroup): the tables -- hsk_shard_flush included -- stay exactly what they were before the first overflowing
 * step, however late the host reads the status word.  Such a state cannot be trained further: rebuild it with larger
 * capacities.
 * ------------------------------------------------------------------------------------------ */
enum { HSK_STATUS_SHARD_OVERFLOW = 4 };

typedef struct hsk_bprmf_shard {
  hsk_bprmf_state base;       /* LOCAL shards; base.max_batch >= max(world*batch, C), base.max_cols >= n_neg + 1;
                                 base.csr_* / coo_* are the GLOBAL training interactions (global user / item ids) */
  int32_t world, rank;
  int64_t n_users_global, n_items_global;
  int64_t item_lo;            /* first global item id of the local shard (base.n_items rows) */
  int64_t capacity;           /* C: user slots per owner rank */
  int64_t entry_cap;          /* kept (positive, item) entries per rank and step, <= world*batch*(n_neg+1) */
  void* shard_ws;             /* device scratch, hsk_shard_workspace_bytes() bytes, 256-byte aligned */
  int64_t shard_ws_bytes;
  float* rows_send;           /* [C, D]          all_gather input */
  float* rows_all;            /* [world*C, D]    all_gather output */
  float* dU_all;              /* [world*C, D]    reduce_scatter input (zero-initialised by the caller) */
  float* grads_mine;          /* [C, D]          reduce_scatter output */
  float* s0;                  /* [world*batch]   bpr: all_reduce in place; bce: the owners' scores, not exchanged */
  float* gsum;                /* [world*batch]   bpr: all_reduce in place */
  float* ssm_send;            /* [3, world*batch]        sampled softmax only (else may be NULL): all_gather input */
  float* ssm_all;             /* [world, 3, world*batch] sampled softmax only: all_gather output */
  /* ---- LIBRARY SCRATCH: zero-initialise, never write ----------------------------------------------------------- */
  int64_t cur_batch, cur_cols; /* shape of the step in flight (0: none) */
  int32_t cur_set;             /* buffer set of the step in flight */
  int32_t phase;               /* next expected phase */
} hsk_bprmf_shard;

int64_t hsk_shard_workspace_bytes(int64_t max_batch, int64_t max_cols, int64_t capacity, int64_t entry_cap);
/* size of base.workspace with base.ws_sharded = 1 (no [max_batch, dim] row buffers) */
int64_t hsk_shard_base_workspace_bytes(int64_t n_users, int64_t n_items, int64_t dim, int64_t max_batch,
                                       int64_t max_cols);
int hsk_shard_init(hsk_bprmf_shard* sh, hsk_stream_t stream);   /* after hsk_bprmf_init_workspace(&sh->base) */
int hsk_shard_prepare(hsk_bprmf_shard* sh, const int64_t* order, int64_t start_global, int64_t batch, int64_t n_neg,
                      int32_t set, hsk_stream_t stream);
/* a batch prepared in `set` that will not be trained on (a wrong guess of the next batch): releases its owner map */
int hsk_shard_discard(hsk_bprmf_shard* sh, int32_t set, hsk_stream_t stream);
int hsk_shard_pack(hsk_bprmf_shard* sh, int64_t batch, int64_t n_neg, int32_t set, hsk_stream_t stream);
int hsk_shard_pos_scores(hsk_bprmf_shard* sh, hsk_stream_t stream);
int hsk_shard_forward(hsk_bprmf_shard* sh, hsk_stream_t stream);
int hsk_shard_pos_fix(hsk_bprmf_shard* sh, hsk_stream_t stream);
int hsk_shard_apply_items(hsk_bprmf_shard* sh, hsk_stream_t stream);
int hsk_shard_apply_users(hsk_bprmf_shard* sh, hsk_stream_t stream);
/* bring every local user (and lazily updated item) row up to date (before evaluation / gathering the tables) */
int hsk_shard_flush(hsk_bprmf_shard* sh, hsk_stream_t stream);
/* debug / parity: the kept entries of the batch in flight or last prepared in `set`: counts per positive
 * offs [world*batch + 1], local item ids items [entry_cap], user ids u [world*batch] (device int32 arrays) */
int hsk_shard_last_batch(const hsk_bprmf_shard* sh, int32_t set, int64_t batch, int32_t* offs_out, int32_t* items_out,
                         int32_t* u_out, hsk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The sharded step issued from ONE C call, its collectives called through a table of function pointers
 * (csrc/hsk_rccl.inc).  The phase functions above leave the collectives to the caller (hassaku_amd/dist.py:
 * torch.distributed), which costs ten host dispatches per step; hsk_shard_step enqueues the same kernels and the
 * collectives between them -- the gradients' way home on a communication stream, so that the item pass overlaps it.
 * Same results as the phased sequence.  Replaces, with everything above, nn.DataParallel (train/trainer.py:38-41).
 *
 * The table: float32 collectives over the job's ranks, enqueued on `stream` in stream order, 0 on success.
 *   all_gather(send [count] -> recv [world*count])   all_reduce_sum(buf [count], in place)
 *   reduce_scatter_sum(send [world*count] -> recv [count] = sum over ranks of their send[rank*count ..])
 * Built in: RCCL (loaded at run time by dlopen; hsk_rccl_unique_id) and a HOST-STAGED table for ranks that are
 * processes sharing ONE GPU (hsk_hostcoll_unique_id: shared memory + host functions on the stream; a test backend --
 * RCCL refuses two ranks on one device -- so that hsk_shard_step itself runs at world > 1 on a one-GPU box).
 *   hsk_rccl_available      1 if librccl could be loaded
 *   hsk_rccl_unique_id      rank 0: 128 bytes (host) that the caller hands to every rank by any transport
 *   hsk_hostcoll_unique_id  rank 0: the same for the host-staged table; max_floats = the largest per-rank send of one
 *                           collective (world * capacity * dim for the step)
 *   hsk_shard_rt_create     collective: every rank, same 128 bytes (either kind), its GPU current -> runtime handle
 *                           (the collectives, side / communication streams, the batch prepared a step ahead), or NULL
 *   hsk_shard_rt_create_with  the same around a caller-provided table (ctx is owned by the handle from then on)
 *   hsk_shard_rt_backend    "rccl", "host-staged" or the injected table's name
 *   hsk_shard_step          one global step; next_start >= 0 names the following call's batch (prepared a step ahead;
 *                           next_batch <= 0: same size); replaces pack .. apply_users of the phased protocol.  Every
 *                           argument is validated before anything is enqueued; a later failure joins the forked streams
 *                           back and leaves the state idle (phase, prepared batch dropped)
 *   hsk_shard_rt_flush      drops a prepared batch that was never trained on, then hsk_shard_flush
 *   hsk_shard_rt_cur_set / _discard_prefetch   debug / parity (which buffer set the last step used)
 * ------------------------------------------------------------------------------------------ */
typedef struct hsk_collectives {
  void* ctx;
  int (*all_gather)(void* ctx, const float* send, float* recv, int64_t count, hsk_stream_t stream);
  int (*all_reduce_sum)(void* ctx, float* buf, int64_t count, hsk_stream_t stream);
  int (*reduce_scatter_sum)(void* ctx, const float* send, float* recv, int64_t count, hsk_stream_t stream);
  void (*destroy)(void* ctx);   /* may be NULL */
  const char* name;             /* static string */
} hsk_collectives;

int hsk_rccl_available(void);
int hsk_rccl_unique_id(void* id128);
int hsk_hostcoll_unique_id(int64_t max_floats, void* id128);
void* hsk_shard_rt_create(int32_t world, int32_t rank, const void* id128);
void* hsk_shard_rt_create_with(int32_t world, int32_t rank, const hsk_collectives* coll);
const char* hsk_shard_rt_backend(const void* rt);
void hsk_shard_rt_destroy(void* rt);
int hsk_shard_step(hsk_bprmf_shard* sh, void* rt, const int64_t* order, int64_t start_global, int64_t batch,
                   int64_t n_neg, int64_t next_start, int64_t next_batch, hsk_stream_t stream);
int hsk_shard_rt_flush(hsk_bprmf_shard* sh, void* rt, hsk_stream_t stream);
int hsk_shard_rt_cur_set(const void* rt);
int hsk_shard_rt_discard_prefetch(hsk_bprmf_shard* sh, void* rt, hsk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Synthetic training interactions generated straight into HBM (BASELINE configs[4]: 100 M users x 10 M items; no CSV
 * is ever written).  Stands where TrainRecDataset._prepare_data builds the COO iteration matrix and the CSR sampling
 * matrix from listening_history_train.csv (data/dataset.py:120-131): same arrays, same invariants (CSR rows sorted and
 * duplicate-free; the COO lists every interaction once).  The COO order is the CSR order: coo_item == csr_indices.
 * A pure function of (seed, user id) -- every rank generates identical arrays; law in csrc/hsk_synth.hip.
 *   hsk_synth_degrees   indptr[0] = 0, indptr[u+1] = deg(u) in [deg_min, deg_min + deg_span); the caller turns the
 *                       counts into offsets by an inclusive scan of indptr[1..] (in place)
 *   hsk_synth_fill      csr_indices[indptr[u] .. indptr[u+1]) = the user's items, coo_user likewise = u (may be NULL);
 *                       deg_max = deg_min + deg_span - 1 <= min(64, n_items); skew = 1 (uniform items), 2, 3 (popular
 *                       low ids: item = floor(I * x^skew), x stratified-uniform)
 * ------------------------------------------------------------------------------------------ */
int hsk_synth_degrees(int64_t n_users, int32_t deg_min, int32_t deg_span, uint64_t seed, int64_t* indptr,
                      hsk_stream_t stream);
int hsk_synth_fill(int64_t n_users, int64_t n_items, int32_t deg_max, int32_t skew, uint64_t seed,
                   const int64_t* indptr, int32_t* csr_indices, int32_t* coo_user, hsk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Full-catalogue evaluation (eval/eval.py:237-253, eval/eval.py:54-99, eval/metrics.py:4-105)
 * ------------------------------------------------------------------------------------------ */

/*
 * scores[r, j] = <user_emb[u[r]], item_emb[item_begin+j]> + biases, j in [0,item_count), fp32
 * (exact-fp32 MFMA); entries (u[r], item) present in the exclude CSR get -inf; then the k best per
 * row are returned sorted by (score desc, item id asc).  out_idx holds GLOBAL item ids.
 * scores_ws: [n_rows, item_count] floats of scratch (also the masked score matrix on return).
 * Only rows [item_begin, item_begin + item_count) of item_emb / item_bias are dereferenced: a rank that holds just
 * that range (item-sharded tables) passes shard - item_begin*dim (resp. shard_bias - item_begin) as the base.
 * Replaces get_item_representations(arange(I)) + combine_user_item_representations + mask +
 * logits.topk (eval/eval.py:240-251, :63) for one item shard.
 */
int hsk_mf_eval_topk(const float* user_emb, const float* item_emb, const float* item_bias,
                     const float* user_bias, const float* global_bias,
                     int64_t n_users, int64_t n_items, int64_t dim,
                     const int64_t* u_idx, int64_t n_rows,
                     int64_t item_begin, int64_t item_count,
                     const int64_t* excl_indptr, const int32_t* excl_indices,
                     int64_t k, float* scores_ws, float* out_vals, int32_t* out_idx,
                     int32_t* status, hsk_stream_t stream);

/*
 * The same call with scratch for the operands' pieces (hsk_eval_set_arith forms 1 and 2): given
 * planes_ws of >= hsk_mf_eval_planes_bytes(n_rows, item_count, dim) bytes (256-byte aligned), the rows are cut into
 * their pieces once up front instead of once per tile inside the GEMM loop -- in form 1 the same bits out; form 2 (fp16
 * pairs) exists only on such pre-cut operands.  planes_ws NULL (or too small, or dim % 4 != 0): exactly
 * hsk_mf_eval_topk, which runs form 1 when form 2 is set.
 */
int64_t hsk_mf_eval_planes_bytes(int64_t n_rows, int64_t item_count, int64_t dim);
int hsk_mf_eval_topk_planes(const float* user_emb, const float* item_emb, const float* item_bias,
                            const float* user_bias, const float* global_bias,
                            int64_t n_users, int64_t n_items, int64_t dim,
                            const int64_t* u_idx, int64_t n_rows,
                            int64_t item_begin, int64_t item_count,
                            const int64_t* excl_indptr, const int32_t* excl_indices,
                            int64_t k, float* scores_ws, void* planes_ws, int64_t planes_bytes,
                            float* out_vals, int32_t* out_idx,
                            int32_t* status, hsk_stream_t stream);

/*
 * The same result WITHOUT the score matrix: the top-k selection runs inside the score GEMM (per-row thresholds and
 * candidate lists, csrc/hsk_eval_fused.hip), so nothing of size n_rows x item_count is ever written -- at 131 072
 * items that matrix is 1 GB per 2048 users.  k <= 128.  ws: hsk_mf_eval_fused_ws_bytes(n_rows, item_count, k) bytes of
 * device scratch, 256-byte aligned.  out_vals / out_idx exactly as hsk_mf_eval_topk returns them.
 * hsk_mf_eval_fused_ws_bytes_dim(..., dim) is the larger size that also holds the pieces of both operands
 * (hsk_eval_set_arith forms 1 and 2): given that much, the call splits the rows once up front instead of once per
 * tile inside the GEMM loop -- form 1: same bits out, less work in the loop; form 2 needs it (without, form 1 runs).
 * Either size is accepted.
 */
int64_t hsk_mf_eval_fused_ws_bytes(int64_t n_rows, int64_t item_count, int64_t k);
int64_t hsk_mf_eval_fused_ws_bytes_dim(int64_t n_rows, int64_t item_count, int64_t k, int64_t dim);
int hsk_mf_eval_topk_fused(const float* user_emb, const float* item_emb, const float* item_bias,
                           const float* user_bias, const float* global_bias,
                           int64_t n_users, int64_t n_items, int64_t dim,
                           const int64_t* u_idx, int64_t n_rows,
                           int64_t item_begin, int64_t item_count,
                           const int64_t* excl_indptr, const int32_t* excl_indices,
                           int64_t k, void* ws, int64_t ws_bytes, float* out_vals, int32_t* out_idx,
                           int32_t* status, hsk_stream_t stream);

/* top-k of each row of a dense [rows, cols] fp32 matrix (leading dimension ld); indices int64
 * (torch.topk dtype); order (value desc, index asc).  Replaces logits.topk(k) (eval/eval.py:63). */
int hsk_topk_dense(const float* logits, int64_t rows, int64_t cols, int64_t ld, int64_t k,
                   float* out_vals, int64_t* out_idx, hsk_stream_t stream);

/* merge n_parts candidate lists [n_parts, rows, k] (e.g. all-gathered item shards) into the global
 * top-k per row, same ordering rule. */
int hsk_topk_merge(const float* vals, const int32_t* idx, int64_t n_parts, int64_t rows, int64_t k,
                   float* out_vals, int32_t* out_idx, hsk_stream_t stream);

/*
 * Per-user precision@k, recall@k, ndcg@k for each k in ks (host array, descending or any order)
 * from a ranked list topk_idx [n_rows, k_max] and the ground-truth CSR (sorted rows) of the split:
 * out[r, 3*t+0..2] = precision, recall, ndcg at ks[t].  Semantics of eval/metrics.py:4-105
 * (recall and ndcg are 0 for users without ground truth; ndcg clamped to <= 1).
 */
int hsk_rank_metrics(const int32_t* topk_idx, int64_t n_rows, int64_t k_max,
                     const int64_t* u_idx, int64_t n_users,
                     const int64_t* label_indptr, const int32_t* label_indices,
                     const int32_t* ks, int32_t n_ks, float* out, hsk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HASSAKU_HIP_H */
