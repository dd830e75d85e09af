// hsk_fused.hip -- the fused BPR-MF AdamW training step of the C ABI (hsk_bprmf_*): workspace layout,
// per-stage HIP-event timing, and the launch sequence
//
//   prep     loader batch -> int32 copies   |   device batch: COO gather + uniform rejection sampler
//   sort     hist -> scan -> scatter -> bucket sort   (entries grouped by item, deterministic)
//   fwd      k_fwd_ugrad      gather + scores + BPR + user-row gradient      (the roofline kernel)
//   item     k_item_update    item-major gradient reduction + AdamW on every item row
//   user     k_user_update[_lazy]  AdamW on the user table (dense sweep, or exact lazy catch-up)
//   finish   loss reduction, global bias
//
// Reference semantics: one iteration of the loop at train/trainer.py:128-148 of the reference.
#include <hip/hip_ext.h>
#include "hsk_sampler.h"
#include "hsk_sort.h"
#include "hsk_step_kernels.h"
#include "hsk_item_sliced.h"
#include "hsk_fwd_small.h"
#include "hsk_fwd_part.h"
#include <stdlib.h>

#include <algorithm>
#include <utility>
#include <vector>

#ifndef HSK_FWD_R
#define HSK_FWD_R 4   // item rows per buffer of the forward kernels (two buffers in flight), rows of fewer than 16 floats per lane
#endif
#ifndef HSK_FWD_PART_R
#define HSK_FWD_PART_R 3   // ... of the item-partitioned forward: with the riding preparation phases in the kernel, four rows
                           // per buffer need 129 registers (three waves per SIMD); three rows: 114, and the same step time
                           // (188.3 against 188 us, measured in round 3 before the phases rode along)
#endif
#define HSK_ADAM_TAB_LEN 65536   // per-step (step_size, bc2_sqrt) table for the lazy replay
#define HSK_FLUSH_NEVER (1 << 30) // cadence of a table whose periodic sweep never pays (an explicit flush still sweeps)

// =============================================================================================
// item-partitioned forward (hsk_fwd_part.h): how many partitions for this table and batch
// =============================================================================================
// Pays when the item table is larger than an XCD's 4 MB L2 but a partition of it (mostly) fits, the batch fills the
// chip, and a (positive, partition) unit still has a few rows to keep in flight.  Pure function of its arguments: the
// sampler that buckets a batch's negatives and the step that trains on it must agree.  HSK_FWD_PARTS=1 turns it off,
// =2/4/8 forces a partition count (where the shape allows partitions at all).
#define HSK_PART_MAX 8
// Grouped preparation (small batches, replayed graphs): the batches of G consecutive steps are sampled and sorted by ONE
// launch each (k_prep_sample_group, k_sort_lds over G workgroups) -- at B = 128 a step is four ~5 us kernels, two of which
// only prepare the next batch.  The per-batch buffers of a set then hold G batches; eager steps use slot 0.
#define HSK_GROUP_MAX 8
static int hsk_group_rule(int64_t n_items, int64_t dim, int64_t batch, int64_t n_cols) {
  static const int env = getenv("HSK_GROUP") ? atoi(getenv("HSK_GROUP")) : HSK_GROUP_MAX;
  if (env <= 1 || dim % 2 != 0 || batch > 1024 || !hsk_sort_lds_fits(n_items, batch * n_cols)) return 1;
  return std::min(env, HSK_GROUP_MAX);
}

static inline int64_t hsk_part_cols(int64_t n_cols, int n_part) { return n_cols + (n_part > 1 ? n_part - 1 : 0); }
static int hsk_part_rule(int64_t n_items, int64_t dim, int64_t batch, int64_t n_neg, bool lazy_items) {
  static const int env = getenv("HSK_FWD_PARTS") ? atoi(getenv("HSK_FWD_PARTS")) : 0;
  if (env == 1 || lazy_items) return 1;
  // (the partitioned kernel exists for rows of 64 lanes x 4 floats x {1, 2, 4, 8} whole chunks: 256, 512, 1024, 2048 --
  // 768 passed the old `dim % 256 == 0` test and then found no kernel: tools/stress_pipeline.py)
  if ((dim != 256 && dim != 512 && dim != 1024 && dim != 2048) || batch < 2048 || n_neg > 256 || n_items < 64) return 1;
  int P;
  if (env == 2 || env == 4 || env == 8) {
    P = env;
  } else {
    const double mb = (double)n_items * (double)dim * 4.0 / (1024.0 * 1024.0);
    // measured per step at D = 512, N = 100, B = 4096 (rule vs no partitions): 7.6 MB (P = 2) 180 vs 182 us, 21.9 MB
    // (P = 4) 199 vs 220, 41 MB (P = 8) 250 vs 259, 61 MB (P = 8) 298 vs 295: beyond ~48 MB the partial rows cost more
    // than the L2 share gains
    P = mb <= 4.5 ? 1 : mb <= 12.0 ? 2 : mb <= 24.0 ? 4 : mb <= 48.0 ? 8 : 1;
  }
  while (P > 1 && n_neg < 8 * P) P >>= 1;
  return P;
}
// ... for a state: no partitions under lazy item AdamW, nor under sampled softmax (the partitioned kernel carries the bpr and
// bce epilogues only -- a sampled-softmax run with a large batch on a mid-size item table ran the bpr one:
// tests/stress_step.py)
static int hsk_part_rule_st(const hsk_bprmf_state* st, int64_t batch, int64_t n_neg) {
  return hsk_part_rule(st->n_items, st->dim, batch, n_neg, st->lazy_items != 0 || st->loss_kind == HSK_LOSS_SSM);
}

// =============================================================================================
// workspace carving
// =============================================================================================
struct hsk_ws {
  int* u32;
  int* it32;
  float* g_s;
  int2* perm1;
  int* perm;
  int* hist;
  int* btot;
  int* bstart;
  int* offsets;
  int* owner;
  int* cnt;
  int* last_step;
  float* dUb;
  float *ucur, *mcur, *vcur;   // [max_batch, D] each: the batch's user rows (and, for rows that were behind, their
                               // replayed moments) as the forward saw them (lazy user AdamW)
  double* loss_b;
  float2* adam_tab;
  hsk_step_desc* desc;     // device-resident step descriptor of a replayed graph (hsk_rows.h)
  int *dupcnt, *duplist;   // per owner entry: how many / which further entries name the same user
  int* last_step_i;        // lazy item AdamW: steps already applied to each item row
  int *stamp, *stamp_b;    // per buffer set: stamp[u] = step the set's batch is trained on (for its users)
  int* claim;              // ahead-of-time catch-up: last step at which a row was claimed
  int *touched, *n_touched, *touched_b, *n_touched_b;   // per buffer set: items with entries (compact), their count
  int* pend;               // lazy item AdamW: per entry of the step's touched list, the step its moments stand at
  // item-partitioned forward: dUb holds n_part_max partial rows per batch position, loss_b as many planes; the batch rows
  // have n_part - 1 extra columns
  int n_part_max;
  // grouped preparation: slots per set and the distance (in elements) between the slots of a per-batch buffer
  int group;
  int64_t gs_batch, gs_ent, gs_items, gs_users;
  // second set of the per-batch buffers: the next batch is sampled and sorted into it while this one trains
  int *u32_b, *it32_b, *perm_b, *hist_b, *btot_b, *bstart_b, *offsets_b, *owner_b, *cnt_b;
  int2* perm1_b;
  // third set (single-GPU step only): the in-launch preparation pipeline works on the batches of THREE steps at a time
  int *u32_c, *it32_c, *perm_c, *hist_c, *btot_c, *bstart_c, *offsets_c, *owner_c, *cnt_c, *stamp_c;
  int2* perm1_c;
  int64_t total;
};

// view with the per-batch buffers of `set` in the primary slots
static inline hsk_ws hsk_select(const hsk_ws& w0, int set, int slot = 0) {
  hsk_ws w = w0;
  if (slot > 0) {   // slot `slot` of both sets' per-batch buffers (grouped preparation)
    for (int** p : {&w.u32, &w.u32_b}) *p += slot * w.gs_batch;   // (grouped preparation uses sets 0 / 1 only)
    for (int** p : {&w.it32, &w.it32_b, &w.perm, &w.perm_b}) *p += slot * w.gs_ent;
    for (int** p : {&w.offsets, &w.offsets_b}) *p += slot * w.gs_items;
    for (int** p : {&w.owner, &w.owner_b, &w.cnt, &w.cnt_b, &w.stamp, &w.stamp_b}) *p += slot * w.gs_users;
  }
  if (set == 0) return w;
  hsk_ws r = w;
  if (set == 2) {   // (pipeline only; the lazy-item lists have no third set: the pipeline never runs with lazy items)
    std::swap(r.u32, r.u32_c);
    std::swap(r.it32, r.it32_c);
    std::swap(r.perm1, r.perm1_c);
    std::swap(r.perm, r.perm_c);
    std::swap(r.hist, r.hist_c);
    std::swap(r.btot, r.btot_c);
    std::swap(r.bstart, r.bstart_c);
    std::swap(r.offsets, r.offsets_c);
    std::swap(r.owner, r.owner_c);
    std::swap(r.cnt, r.cnt_c);
    std::swap(r.stamp, r.stamp_c);
    return r;
  }
  std::swap(r.u32, r.u32_b);
  std::swap(r.it32, r.it32_b);
  std::swap(r.perm1, r.perm1_b);
  std::swap(r.perm, r.perm_b);
  std::swap(r.hist, r.hist_b);
  std::swap(r.btot, r.btot_b);
  std::swap(r.bstart, r.bstart_b);
  std::swap(r.offsets, r.offsets_b);
  std::swap(r.owner, r.owner_b);
  std::swap(r.cnt, r.cnt_b);
  std::swap(r.stamp, r.stamp_b);
  std::swap(r.touched, r.touched_b);
  std::swap(r.n_touched, r.n_touched_b);
  return r;
}

// sharded: the layout of a state with ws_sharded = 1 -- the sharded step keeps the batch's user rows and gradient rows in
// its exchange buffers (rows_all / dU_all), so the [max_batch, D] row buffers, the partial-row planes and the
// ahead-of-time bookkeeping of the single-GPU step are not carved (several hundred MB per rank at W = 8, D = 512)
static hsk_ws hsk_carve(void* base, int64_t n_users, int64_t n_items, int64_t dim, int64_t max_batch,
                        int64_t max_cols, bool sharded = false) {
  hsk_ws w;
  char* p = (char*)base;
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    char* r = p ? p + off : nullptr;
    off += hsk_align_up(bytes, 256);
    return r;
  };
  w.n_part_max = sharded ? 1 : hsk_part_rule(n_items, dim, max_batch, std::min<int64_t>(max_cols - 1, 256), false);   // an upper bound
  const int64_t row_elems = sharded ? 0 : max_batch * dim;   // one [max_batch, D] row buffer
  const int64_t ent = max_batch * (max_cols + w.n_part_max - 1);   // partitioned rows: the positive n_part times
  const int64_t hist_elems = hsk_sort_hist_elems(n_items, ent);
  const int G = hsk_group_rule(n_items, dim, max_batch, max_cols);
  w.group = G;
  w.gs_batch = max_batch;
  w.gs_ent = ent;
  w.gs_items = n_items + 1;
  w.gs_users = n_users;
  w.u32 = (int*)take(G * max_batch * 4);
  w.it32 = (int*)take(G * ent * 4);
  w.g_s = (float*)take(ent * 4);
  w.perm1 = (int2*)take(ent * 8);
  w.perm = (int*)take(G * ent * 4);
  w.hist = (int*)take((hist_elems > 0 ? hist_elems : 4) * 4);
  w.btot = (int*)take(HSK_SORT_MAX_BUCKETS * 4);
  w.bstart = (int*)take((HSK_SORT_MAX_BUCKETS + 1) * 4);
  w.offsets = (int*)take(G * (n_items + 1) * 4);
  w.owner = (int*)take(G * n_users * 4);
  w.cnt = (int*)take(G * n_users * 4);
  w.last_step = (int*)take(n_users * 4);
  w.dUb = (float*)take(w.n_part_max * row_elems * 4);
  w.ucur = (float*)take(row_elems * 4);
  w.mcur = (float*)take(row_elems * 4);
  w.vcur = (float*)take(row_elems * 4);
  w.loss_b = (double*)take(w.n_part_max * max_batch * 8);
  w.adam_tab = (float2*)take((HSK_ADAM_TAB_LEN + 1) * sizeof(float2));
  w.desc = (hsk_step_desc*)take(256);
  w.dupcnt = (int*)take(max_batch * 4);
  w.duplist = (int*)take(max_batch * HSK_DUP_MAX * 4);
  w.last_step_i = (int*)take(n_items * 4);
  w.stamp = (int*)take(sharded ? 0 : G * n_users * 4);
  w.stamp_b = (int*)take(sharded ? 0 : G * n_users * 4);
  w.claim = (int*)take(sharded ? 0 : n_users * 4);
  w.touched = (int*)take(ent * 4);
  w.touched_b = (int*)take(ent * 4);
  w.n_touched = (int*)take(256);
  w.n_touched_b = (int*)take(256);
  w.pend = (int*)take(ent * 4);
  w.u32_b = (int*)take(G * max_batch * 4);
  w.it32_b = (int*)take(G * ent * 4);
  w.perm1_b = (int2*)take(ent * 8);
  w.perm_b = (int*)take(G * ent * 4);
  w.hist_b = (int*)take((hist_elems > 0 ? hist_elems : 4) * 4);
  w.btot_b = (int*)take(HSK_SORT_MAX_BUCKETS * 4);
  w.bstart_b = (int*)take((HSK_SORT_MAX_BUCKETS + 1) * 4);
  w.offsets_b = (int*)take(G * (n_items + 1) * 4);
  w.owner_b = (int*)take(G * n_users * 4);
  w.cnt_b = (int*)take(G * n_users * 4);
  const int64_t c3 = sharded ? 0 : 1;   // third set: one batch (no grouped slots), not carved for the sharded step
  w.u32_c = (int*)take(c3 * max_batch * 4);
  w.it32_c = (int*)take(c3 * ent * 4);
  w.perm1_c = (int2*)take(c3 * ent * 8);
  w.perm_c = (int*)take(c3 * ent * 4);
  w.hist_c = (int*)take(c3 * (hist_elems > 0 ? hist_elems : 4) * 4);
  w.btot_c = (int*)take(c3 * HSK_SORT_MAX_BUCKETS * 4);
  w.bstart_c = (int*)take(c3 * (HSK_SORT_MAX_BUCKETS + 1) * 4);
  w.offsets_c = (int*)take(c3 * (n_items + 1) * 4);
  w.owner_c = (int*)take(c3 * n_users * 4);
  w.cnt_c = (int*)take(c3 * n_users * 4);
  w.stamp_c = (int*)take(c3 * n_users * 4);
  w.total = off;
  return w;
}

static inline hsk_ws hsk_carve_st(const hsk_bprmf_state* st) {
  return hsk_carve(st->workspace, st->n_users, st->n_items, st->dim, st->max_batch, st->max_cols, st->ws_sharded != 0);
}

extern "C" int64_t hsk_bprmf_workspace_bytes(int64_t n_users, int64_t n_items, int64_t dim, int64_t max_batch,
                                             int64_t max_cols) {
  if (n_users <= 0 || n_items <= 0 || dim <= 0 || max_batch <= 0 || max_cols <= 1) return -1;
  if (hsk_sort_hist_elems(n_items, max_batch * (max_cols + HSK_PART_MAX - 1)) < 0) return -1;
  return hsk_carve(nullptr, n_users, n_items, dim, max_batch, max_cols).total;
}

extern "C" int64_t hsk_shard_base_workspace_bytes(int64_t n_users, int64_t n_items, int64_t dim, int64_t max_batch,
                                                  int64_t max_cols) {
  if (n_users <= 0 || n_items <= 0 || dim <= 0 || max_batch <= 0 || max_cols <= 1) return -1;
  if (hsk_sort_hist_elems(n_items, max_batch * (max_cols + HSK_PART_MAX - 1)) < 0) return -1;
  return hsk_carve(nullptr, n_users, n_items, dim, max_batch, max_cols, true).total;
}

// lazy replay needs the per-step scalars to be constant beyond the table
static bool hsk_adam_tab_saturates(const hsk_bprmf_state* st) {
  const hsk_adamw_consts a = hsk_make_adamw_consts(st->lr, st->beta1, st->beta2, st->eps, st->wd, HSK_ADAM_TAB_LEN, st->opt_kind);
  const hsk_adamw_consts b = hsk_make_adamw_consts(st->lr, st->beta1, st->beta2, st->eps, st->wd, (int64_t)1 << 40);
  return a.step_size == b.step_size && a.bc2_sqrt == b.bc2_sqrt && a.rbc2_sqrt == b.rbc2_sqrt;
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
  HSK_REQUIRE(st->max_batch > 0 && st->max_cols >= 2 && st->max_batch * st->max_cols < 0x7fffffff, HSK_ERR_INVALID,
              "bad max_batch / max_cols");
  HSK_REQUIRE(st->workspace != nullptr, HSK_ERR_INVALID, "workspace is NULL");
  HSK_REQUIRE(st->ws_sharded == 0 || st->ws_sharded == 1, HSK_ERR_INVALID, "ws_sharded must be 0 or 1");
  HSK_REQUIRE(st->flush_every >= 0, HSK_ERR_INVALID, "flush_every must be >= 0");
  const int64_t need = st->ws_sharded
                           ? hsk_shard_base_workspace_bytes(st->n_users, st->n_items, st->dim, st->max_batch, st->max_cols)
                           : hsk_bprmf_workspace_bytes(st->n_users, st->n_items, st->dim, st->max_batch, st->max_cols);
  HSK_REQUIRE(need > 0, HSK_ERR_UNSUPPORTED, "n_items %lld too large for the item sort (max %d per device)",
              (long long)st->n_items, HSK_SORT_MAX_BUCKETS * HSK_SORT_MAX_IPB);
  HSK_REQUIRE(st->workspace_bytes >= need, HSK_ERR_INVALID, "workspace too small: %lld < %lld",
              (long long)st->workspace_bytes, (long long)need);
  HSK_REQUIRE(((uintptr_t)st->workspace & 255) == 0, HSK_ERR_INVALID, "workspace must be 256-byte aligned");
  const int V = (st->dim % 4 == 0) ? 4 : (st->dim % 2 == 0) ? 2 : 1;
  HSK_REQUIRE((((uintptr_t)st->user_emb | (uintptr_t)st->item_emb | (uintptr_t)st->m_user_emb |
                (uintptr_t)st->v_user_emb | (uintptr_t)st->m_item_emb | (uintptr_t)st->v_item_emb) &
               (uintptr_t)(4 * V - 1)) == 0,
              HSK_ERR_INVALID, "tables must be %d-byte aligned", 4 * V);
  HSK_REQUIRE(st->lazy_users == 0 || st->lazy_users == 1, HSK_ERR_INVALID, "lazy_users must be 0 or 1");
  HSK_REQUIRE(st->lazy_users == 0 || hsk_adam_tab_saturates(st), HSK_ERR_UNSUPPORTED,
              "lazy_users needs bias corrections that saturate within %d steps (betas too close to 1)",
              HSK_ADAM_TAB_LEN);
  HSK_REQUIRE(st->step >= 0 && st->step < 0x7ffffff0, HSK_ERR_UNSUPPORTED, "step counter out of range");
  HSK_REQUIRE(st->loss_kind >= HSK_LOSS_BPR && st->loss_kind <= HSK_LOSS_SSM, HSK_ERR_INVALID, "unknown loss_kind %d",
              st->loss_kind);
  HSK_REQUIRE(st->lazy_items == 0 || st->lazy_items == 1, HSK_ERR_INVALID, "lazy_items must be 0 or 1");
  HSK_REQUIRE(st->lazy_items == 0 || (st->dim % 2 == 0 && hsk_adam_tab_saturates(st)), HSK_ERR_UNSUPPORTED,
              "lazy_items needs an even dim and bias corrections that saturate within %d steps", HSK_ADAM_TAB_LEN);
  HSK_REQUIRE(st->opt_kind >= HSK_OPT_ADAMW && st->opt_kind <= HSK_OPT_ADAGRAD, HSK_ERR_INVALID, "unknown opt_kind %d",
              st->opt_kind);
  HSK_REQUIRE((st->alias_prob == nullptr) == (st->alias_idx == nullptr), HSK_ERR_INVALID,
              "alias table needs both alias_prob and alias_idx");
  // the device table of per-step Adam scalars (lazy replay, replayed graphs) was computed from these at init time
  if (st->frozen_valid) {
    const double now[5] = {st->lr, st->beta1, st->beta2, st->eps, st->wd};
    HSK_REQUIRE(memcmp(now, st->frozen_hyper, sizeof(now)) == 0 && st->opt_kind == st->frozen_opt, HSK_ERR_INVALID,
                "lr / betas / eps / wd / optimizer changed after hsk_bprmf_init_workspace: flush, then call "
                "hsk_bprmf_init_workspace again with the new values");
  }
  return HSK_OK;
}

static void hsk_aux_drop_graphs(void* aux);   // (defined with struct hsk_aux below)
static void hsk_aux_forget_pipeline(void* aux);

extern "C" int hsk_bprmf_init_workspace(hsk_bprmf_state* st, hsk_stream_t stream_) {
  HSK_REQUIRE(st != nullptr, HSK_ERR_INVALID, "state is NULL");
  st->frozen_valid = 0;
  int rc = hsk_check_state(st);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  hsk_ws w = hsk_carve_st(st);
  const int64_t GU = (int64_t)w.group * st->n_users;   // grouped preparation: G owner maps / stamps per set
  HSK_HIP(hipMemsetAsync(w.cnt, 0, GU * 4, stream));
  HSK_HIP(hipMemsetAsync(w.cnt_b, 0, GU * 4, stream));
  HSK_HIP(hipMemsetAsync(w.dupcnt, 0, st->max_batch * 4, stream));
  if (!st->ws_sharded) {
    HSK_HIP(hipMemsetAsync(w.stamp, 0, GU * 4, stream));
    HSK_HIP(hipMemsetAsync(w.stamp_b, 0, GU * 4, stream));
    HSK_HIP(hipMemsetAsync(w.claim, 0, st->n_users * 4, stream));
    HSK_HIP(hipMemsetAsync(w.stamp_c, 0, st->n_users * 4, stream));
    HSK_HIP(hipMemsetAsync(w.cnt_c, 0, st->n_users * 4, stream));
    k_fill_i32<<<(unsigned)hsk_ceil_div(st->n_users, 256), 256, 0, stream>>>(w.owner_c, st->n_users, HSK_OWNER_NONE);
  }
  HSK_HIP(hipMemsetAsync(w.n_touched, 0, 4, stream));
  HSK_HIP(hipMemsetAsync(w.n_touched_b, 0, 4, stream));
  k_fill_i32<<<(unsigned)hsk_ceil_div(st->n_items, 256), 256, 0, stream>>>(w.last_step_i, st->n_items, (int)st->step);
  k_fill_i32<<<(unsigned)hsk_ceil_div(GU, 256), 256, 0, stream>>>(w.owner, GU, HSK_OWNER_NONE);
  k_fill_i32<<<(unsigned)hsk_ceil_div(GU, 256), 256, 0, stream>>>(w.owner_b, GU, HSK_OWNER_NONE);
  HSK_LAUNCH_CHECK();
  // rows are current up to the steps already applied (0 for a fresh optimiser)
  k_fill_i32<<<(unsigned)hsk_ceil_div(st->n_users, 256), 256, 0, stream>>>(w.last_step, st->n_users, (int)st->step);
  HSK_LAUNCH_CHECK();
  if (st->loss_out) HSK_HIP(hipMemsetAsync(st->loss_out, 0, 2 * sizeof(double), stream));
  if (st->status) HSK_HIP(hipMemsetAsync(st->status, 0, sizeof(int32_t), stream));
  // per-step scalars, computed on the host by the same routine every step uses (one-time, synchronous)
  std::vector<float2> tab(HSK_ADAM_TAB_LEN + 1);
  tab[0] = make_float2(0.f, 1.f);
  for (int t = 1; t <= HSK_ADAM_TAB_LEN; ++t) {
    const hsk_adamw_consts c = hsk_make_adamw_consts(st->lr, st->beta1, st->beta2, st->eps, st->wd, t, st->opt_kind);
#if HSK_ADAM_IEEE
    tab[t] = make_float2(c.step_size, c.bc2_sqrt);
#else
    tab[t] = make_float2(c.step_size, c.rbc2_sqrt);
#endif
  }
  HSK_HIP(hipMemcpyAsync(w.adam_tab, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice, stream));
  HSK_HIP(hipStreamSynchronize(stream));
  const double now[5] = {st->lr, st->beta1, st->beta2, st->eps, st->wd};
  memcpy(st->frozen_hyper, now, sizeof(now));
  st->frozen_opt = st->opt_kind;
  st->frozen_valid = 1;
  hsk_aux_drop_graphs(st->aux);   // graphs captured for the previous hyper-parameters are stale
  hsk_aux_forget_pipeline(st->aux);   // (the owner maps were just re-initialised: nothing to give back)
  return HSK_OK;
}

// ---------------------------------------------------------------------------------------------
// per-stage HIP-event timing (host-side recorder; events are recorded on the kernels' own stream)
// ---------------------------------------------------------------------------------------------
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
  if (!t || !st->timing_now || !((st->timing_mask >> stage) & 1)) return;
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

// ---------------------------------------------------------------------------------------------
// aux handle: side stream on which the NEXT batch is sampled and sorted while the current one trains
// ---------------------------------------------------------------------------------------------
// prep + item sort (5 short, latency-bound kernels, ~48 us at the ml10m shape) depend on nothing a step writes
// except the owner map they build themselves.  With an aux handle and a hint naming the next batch
// (hsk_bprmf_hint_next) they run on the side stream, into the other set of per-batch buffers, forked after the
// forward kernel (which occupies every wave slot of the chip on its own) so that they overlap the item and user
// passes.  The next call joins on one event.  Results are identical to the inline order: same RNG stream id,
// same sort.
struct hsk_aux {
  hipStream_t side = nullptr;
  hipStream_t cap = nullptr;   // graphs are captured on this stream (the caller's may be the legacy default stream,
                               // which cannot be captured) and replayed on the caller's
  hipEvent_t ev_fork = nullptr, ev_ready = nullptr;
  bool hint_valid = false;
  const int64_t* hint_order = nullptr;
  int64_t hint_start = 0, hint_batch = 0, hint_nneg = 0;
  // the batch that follows the next hsk_bprmf_train_steps run (hsk_bprmf_hint_after_run): hinted by the run's last step
  bool tail_valid = false;
  const int64_t* tail_order = nullptr;
  int64_t tail_start = 0, tail_batch = 0, tail_nneg = 0;
  bool pf_valid = false;  // a prefetched batch sits in set `cur_set ^ 1`
  const int64_t* pf_order = nullptr;
  int64_t pf_start = 0, pf_batch = 0, pf_nneg = 0, pf_step = 0;
  int cur_set = 0;  // buffers of the batch of the latest step
  int cur_slot = 0; // ... and its slot inside the set (grouped preparation; 0 otherwise)
  // the same for hsk_bprmf_last_batch / _last_sort: cur_set is folded back into {0, 1} when the pipeline is reset (a flush),
  // what the debug reads want is where the last step's batch still sits
  int last_set = 0, last_slot = 0;
  bool grouped = false;   // capture of a run with grouped preparation: the steps launch no prefetch of their own
  // graph capture context: while `g_desc` is set the launch sequence is being CAPTURED, not run: kernels take their
  // per-step scalars from the device descriptor + the relative step `g_rel`, launches carry no events of their own
  const hsk_step_desc* g_desc = nullptr;
  int g_rel = 0;
  struct graph_entry {
    hipGraphExec_t exec;
    int64_t n_steps, batch, n_neg;
    int set0, flags;
    hsk_bprmf_state key;   // the state as captured (hsk_graph_key): a replay is only valid for exactly this state
  };
  std::vector<graph_entry> graphs;
  bool graph_broken = false;   // a capture failed once: stay with eager launches
  int64_t graph_launches = 0;  // replayed runs so far (hsk_bprmf_graph_replays)
  // in-launch preparation pipeline (hsk_pipe_*, below): the batches of the current and of the next two steps, one buffer
  // set each; `stage` = the preparation phases already done (or enqueued) for the batch in the slot
  struct pipe_slot {
    bool valid = false;
    const int64_t* order = nullptr;
    int64_t start = 0, batch = 0, n_neg = 0;
    int64_t step = 0;   // the batch is trained on when st->step == step (its RNG stream id)
    int stage = 0;      // HSK_PIPE_*
  } pipe[3];
  bool pipe_active = false;                 // a pipelined step is being issued: hsk_run_step forks no prefetch of its own
  const hsk_ride_fwd* ride_fwd = nullptr;   // what rides in the two launches of that step
  const hsk_ride_item* ride_item = nullptr;
  int64_t tail_count = 0;                   // consecutive batches the caller named behind the next run (tail_*)
  int64_t pipe_steps = 0;                   // steps issued through the pipeline so far (debug / tests)
};
enum { HSK_PIPE_NONE = 0, HSK_PIPE_S = 1, HSK_PIPE_H = 2, HSK_PIPE_R = 3, HSK_PIPE_C = 4, HSK_PIPE_B = 5 };

static void hsk_aux_drop_graphs(void* a_) {
  hsk_aux* a = (hsk_aux*)a_;
  if (!a) return;
  for (auto& g : a->graphs) (void)hipGraphExecDestroy(g.exec);
  a->graphs.clear();
}

static void hsk_aux_forget_pipeline(void* a_) {
  hsk_aux* a = (hsk_aux*)a_;
  if (!a) return;
  for (auto& b : a->pipe) b = hsk_aux::pipe_slot{};
  if (a->cur_set > 1) a->cur_set = 0;
}

extern "C" void hsk_aux_destroy(void* a_) {
  hsk_aux* a = (hsk_aux*)a_;
  if (!a) return;
  if (a->side) (void)hipStreamSynchronize(a->side);
  for (auto& g : a->graphs) (void)hipGraphExecDestroy(g.exec);
  if (a->ev_fork) (void)hipEventDestroy(a->ev_fork);
  if (a->ev_ready) (void)hipEventDestroy(a->ev_ready);
  if (a->side) (void)hipStreamDestroy(a->side);
  if (a->cap) (void)hipStreamDestroy(a->cap);
  delete a;
}

extern "C" void* hsk_aux_create(void) {
  hsk_aux* a = new hsk_aux();
  // Priority of the side stream (measured at the ml10m shape, us/step): lowest 242 (the side chain starves behind
  // the item pass and the next step waits for it), default 224, highest 222: its kernels are short and
  // latency-bound, letting them through costs the item pass less than a late join costs the step.
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  const char* pe = getenv("HSK_SIDE_PRIO");
  const int prio = pe ? atoi(pe) : prio_hi;
  bool ok = hipStreamCreateWithPriority(&a->side, hipStreamNonBlocking, prio) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&a->cap, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreate(&a->ev_fork) == hipSuccess;   // a kernel stop event: must be able to take a timestamp
  ok = ok && hipEventCreateWithFlags(&a->ev_ready, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    hsk_set_error("hsk_aux_create: could not create the side stream / events");
    hsk_aux_destroy(a);
    return nullptr;
  }
  return a;
}

extern "C" int hsk_bprmf_hint_next(hsk_bprmf_state* st, const int64_t* order, int64_t start, int64_t batch,
                                   int64_t n_neg) {
  HSK_REQUIRE(st != nullptr, HSK_ERR_INVALID, "state is NULL");
  hsk_aux* a = (hsk_aux*)st->aux;
  HSK_REQUIRE(a != nullptr, HSK_ERR_INVALID, "hint_next needs an aux handle in the state");
  if (batch <= 0) {
    a->hint_valid = false;
    return HSK_OK;
  }
  HSK_REQUIRE(batch <= st->max_batch && n_neg >= 1 && n_neg + 1 <= st->max_cols, HSK_ERR_INVALID,
              "hint: batch %lld / n_neg %lld outside the workspace limits", (long long)batch, (long long)n_neg);
  HSK_REQUIRE(start >= 0 && start + batch <= st->nnz, HSK_ERR_INVALID, "hint: range [%lld, %lld) outside nnz %lld",
              (long long)start, (long long)(start + batch), (long long)st->nnz);
  a->hint_valid = true;
  a->hint_order = order;
  a->hint_start = start;
  a->hint_batch = batch;
  a->hint_nneg = n_neg;
  return HSK_OK;
}

extern "C" int hsk_bprmf_hint_after_run_n(hsk_bprmf_state* st, const int64_t* order, int64_t start, int64_t batch,
                                          int64_t n_neg, int64_t n_batches) {
  HSK_REQUIRE(st != nullptr, HSK_ERR_INVALID, "state is NULL");
  hsk_aux* a = (hsk_aux*)st->aux;
  HSK_REQUIRE(a != nullptr, HSK_ERR_INVALID, "hint_after_run needs an aux handle in the state");
  a->tail_valid = false;
  a->tail_count = 0;
  if (batch <= 0 || n_batches <= 0) return HSK_OK;
  HSK_REQUIRE(batch <= st->max_batch && n_neg >= 1 && n_neg + 1 <= st->max_cols && start >= 0 &&
                  start + n_batches * batch <= st->nnz,
              HSK_ERR_INVALID, "hint_after_run: batches outside the workspace limits / the interactions");
  a->tail_valid = true;
  a->tail_order = order;
  a->tail_start = start;
  a->tail_batch = batch;
  a->tail_nneg = n_neg;
  a->tail_count = n_batches;
  return HSK_OK;
}

extern "C" int hsk_bprmf_hint_after_run(hsk_bprmf_state* st, const int64_t* order, int64_t start, int64_t batch,
                                        int64_t n_neg) {
  return hsk_bprmf_hint_after_run_n(st, order, start, batch, n_neg, 1);
}

// =============================================================================================
// periodic sweep of the lazily updated tables: how often
// =============================================================================================
// A sweep brings every row of the table up to the current step: rows * D * 24 bytes of traffic (p, m, v read and
// written), ~4.5 TB/s measured (ml10m user table: 192 us).  Without it a row replays all its pending zero-gradient
// steps when it is next touched: VALU work, ~3.3e-13 s per element and pending step (ml10m: 4096 x 512 elements, 17
// pending steps, 12 us).  With a fraction lambda = touched / rows of the rows touched per step and a sweep every F
// steps, the expected pending count of a touched row is (1 - exp(-lambda F)) / lambda, hence per step
//     cost(F) = T_sweep / F + t_replay * (1 - exp(-lambda F)) / lambda,        cost(never) = t_replay / lambda.
// The minimum over a grid of cadences is taken; results do not depend on the cadence (any replay length is exact).
static int hsk_flush_cadence_rule(double rows, double touched, double dim) {
  if (rows <= 0 || touched <= 0) return HSK_FLUSH_NEVER;
  const double t_sweep = rows * dim * 24.0 / 4.5e12;
  const double t_replay = touched * dim * 3.3e-13;
  const double lambda = std::min(1.0, touched / rows);
  int best = HSK_FLUSH_NEVER;
  double best_cost = t_replay / lambda;
  for (double f = 16.0; f <= 8192.0; f *= 1.25) {
    const double F = std::floor(f);
    const double cost = t_sweep / F + t_replay * (1.0 - std::exp(-lambda * F)) / lambda;
    if (cost < 0.97 * best_cost) {   // (a sweep has to pay for itself with some margin)
      best_cost = cost;
      best = (int)F;
    }
  }
  return best;
}

// table 0: users, 1: items; touched: rows of that table a step touches
static int hsk_flush_cadence(const hsk_bprmf_state* st, int table, double touched) {
  if (table == 0 ? !st->lazy_users : !st->lazy_items) return HSK_FLUSH_NEVER;
  static const int env = getenv("HSK_FLUSH_EVERY") ? atoi(getenv("HSK_FLUSH_EVERY")) : 0;   // experiments
  if (st->flush_every > 0) return st->flush_every;
  if (env > 0) return env;
  const int rule = hsk_flush_cadence_rule((double)(table == 0 ? st->n_users : st->n_items), touched, (double)st->dim);
  // The rule prices the replay by its THROUGHPUT.  A step that touches few rows is a chain of latencies instead: it waits
  // for the ONE row with the longest backlog (the maximum over the batch's rows, ~ln(rows) times the mean), replayed
  // step by step by a single wave -- ml1m shape (B = 128, U = 6040): 23.7 us per step with sweeps every 64 steps, 50.5
  // without any.  There the sweep's job is to bound that chain: every 64 steps at most, as measured.
  return touched < 2048.0 ? std::min(rule, 64) : rule;
}

// distinct rows among `entries` uniform draws from a table of `rows` rows
static double hsk_touched_rows(double rows, double entries) { return rows * (1.0 - std::exp(-entries / rows)); }

extern "C" int32_t hsk_bprmf_flush_cadence(const hsk_bprmf_state* st, int32_t table, int64_t touched_rows) {
  if (!st || table < 0 || table > 1 || touched_rows <= 0) return -1;
  return hsk_flush_cadence(st, table, (double)touched_rows);
}

// =============================================================================================
// launch sequence
// =============================================================================================
// which: bit 0 = user table, bit 1 = item table (only the lazily updated ones are swept)
// g_poison: sharded step only (hsk_guard_skip) -- a run that skipped a step for a capacity overflow sweeps nothing
static int hsk_launch_flush(hsk_bprmf_state* st, const hsk_ws& w, hipStream_t stream, int which = 3,
                            const int* g_poison = nullptr) {
  const int U = (int)st->n_users, D = (int)st->dim;
  if (st->step == 0) return HSK_OK;
  const hsk_adamw_consts c = hsk_make_adamw_consts(st->lr, st->beta1, st->beta2, st->eps, st->wd, st->step, st->opt_kind);
#define HSK_FLUSH(VV, GEN)                                                                                       \
  k_user_flush<VV, GEN><<<(unsigned)U, 256, 0, stream>>>(st->user_emb, st->m_user_emb, st->v_user_emb, st->user_bias, \
                                                         st->m_user_bias, st->v_user_bias, w.last_step, U, D,         \
                                                         (int)st->step, c, w.adam_tab, HSK_ADAM_TAB_LEN, g_poison)
  const bool gen = st->opt_kind != HSK_OPT_ADAMW;
  // one wave per row (k_row_flush_wave) wherever a row fits a wave's registers; HSK_FLUSH_WAVE=0: workgroup per row
  static const int wave_on = getenv("HSK_FLUSH_WAVE") ? atoi(getenv("HSK_FLUSH_WAVE")) : 1;
  const bool wave = wave_on && D <= 64 * 8 * ((D % 4 == 0) ? 4 : (D % 2 == 0) ? 2 : 1);
  auto wave_sweep = [&](float* P, float* M, float* V_, float* Pb, float* Mb, float* Vb, int* last, int n_rows) {
    return hsk_dispatch_dim(D, [&](auto v_, auto n_, auto f_) {
      constexpr int V = decltype(v_)::value, NCH = decltype(n_)::value;
      constexpr bool FULL = decltype(f_)::value;
      const unsigned grid = (unsigned)((n_rows + 3) / 4);
      if (gen)
        k_row_flush_wave<V, NCH, FULL, true><<<grid, 256, 0, stream>>>(P, M, V_, Pb, Mb, Vb, last, n_rows, D, (int)st->step, c,
                                                                     w.adam_tab, HSK_ADAM_TAB_LEN, g_poison);
      else
        k_row_flush_wave<V, NCH, FULL, false><<<grid, 256, 0, stream>>>(P, M, V_, Pb, Mb, Vb, last, n_rows, D, (int)st->step, c,
                                                                      w.adam_tab, HSK_ADAM_TAB_LEN, g_poison);
      return HSK_OK;
    });
  };
  if (wave) {
    if (st->lazy_users && (which & 1)) {
      const int rc = wave_sweep(st->user_emb, st->m_user_emb, st->v_user_emb, st->user_bias, st->m_user_bias, st->v_user_bias,
                                w.last_step, U);
      if (rc != HSK_OK) return rc;
    }
    if (st->lazy_items && (which & 2)) {
      const int rc = wave_sweep(st->item_emb, st->m_item_emb, st->v_item_emb, st->item_bias, st->m_item_bias, st->v_item_bias,
                                w.last_step_i, (int)st->n_items);
      if (rc != HSK_OK) return rc;
    }
    HSK_LAUNCH_CHECK();
    return HSK_OK;
  }
  if (st->lazy_users && (which & 1)) {
    if (D % 2 == 0) {
      if (gen) HSK_FLUSH(2, true); else HSK_FLUSH(2, false);
    } else {
      if (gen) HSK_FLUSH(1, true); else HSK_FLUSH(1, false);
    }
  }
#undef HSK_FLUSH
  if (st->lazy_items && (which & 2)) {   // the same sweep over the item table (even dim guaranteed by hsk_check_state)
    const int I = (int)st->n_items;
    if (gen)
      k_user_flush<2, true><<<(unsigned)I, 256, 0, stream>>>(st->item_emb, st->m_item_emb, st->v_item_emb, st->item_bias,
                                                            st->m_item_bias, st->v_item_bias, w.last_step_i, I, D,
                                                            (int)st->step, c, w.adam_tab, HSK_ADAM_TAB_LEN, g_poison);
    else
      k_user_flush<2, false><<<(unsigned)I, 256, 0, stream>>>(st->item_emb, st->m_item_emb, st->v_item_emb, st->item_bias,
                                                             st->m_item_bias, st->v_item_bias, w.last_step_i, I, D,
                                                             (int)st->step, c, w.adam_tab, HSK_ADAM_TAB_LEN, g_poison);
  }
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

// the periodic sweeps due after the steps (step_before, st->step]; user_rows / entries: what one step touches
static int hsk_periodic_flush(hsk_bprmf_state* st, const hsk_ws& w, hipStream_t stream, int64_t step_before,
                              double user_rows, double entries, const int* g_poison = nullptr) {
  int which = 0;
  if (st->lazy_users) {
    const int64_t F = hsk_flush_cadence(st, 0, user_rows);
    if (st->step / F != step_before / F) which |= 1;
  }
  if (st->lazy_items) {
    const int64_t F = hsk_flush_cadence(st, 1, hsk_touched_rows((double)st->n_items, entries));
    if (st->step / F != step_before / F) which |= 2;
  }
  if (!which) return HSK_OK;
  int rc = HSK_OK;
  HSK_STAGE(HSK_STAGE_USER, rc = hsk_launch_flush(st, w, stream, which, g_poison));
  return rc;
}

// Item-major pass (gradient reduction [+ AdamW]).  Even D: the D-sliced, XCD-affine kernel with the widest slice the
// alignment allows (256 floats per slice at D % 4 == 0: two slices at D=512, each served by 4 XCDs whose L2 then
// holds 4096 x 1 KB of user rows; measured 72 us against 93 us for whole rows).  Odd D: whole-row kernel.
// `Urows` / `urow_index`: where the batch's user rows live (the table + u32, or the exchange buffer + slot_of_b).
// `ua` (optional, APPLY only): the owners' user-row update rides in the same launch (k_item_user), see hsk_item_sliced.h.
template <int V, int NCH, bool FULL, int R, bool APPLY>
static void hsk_launch_item_pass(const hsk_bprmf_state* st, const hsk_ws& w, const float* Urows, const int* urow_index,
                                 int K, const hsk_adamw_consts& c, float* gI_out, float* gIb_out, hipStream_t stream,
                                 int64_t n_entries = 0, const hsk_user_lazy_args* ua = nullptr,
                                 const hsk_ahead_args* aa = nullptr, int n_part = 1, const int* g_ovf = nullptr,
                                 const int* g_poison = nullptr, const hsk_ride_item* ri = nullptr) {
  // g_ovf / g_poison: the sharded step's overflow guard (hsk_guard_skip); its item pass never carries `ua`
  const bool part = n_part > 1;
  const int I = (int)st->n_items, D = (int)st->dim;
  if (D % 2 != 0) {
    k_item_update<V, NCH, FULL, R, APPLY><<<(unsigned)hsk_ceil_div(I, 4), 256, 0, stream>>>(
        Urows, st->item_emb, st->item_bias, st->m_item_emb, st->v_item_emb, st->m_item_bias, st->v_item_bias,
        urow_index, w.g_s, w.perm, w.offsets, I, K, D, c, gI_out, gIb_out, g_ovf, g_poison);
    return;
  }
  // slice width: 64 lanes x VS floats, the widest vector the row alignment allows (narrower slices and more items per
  // wave were measured slower: 112 / 84-88 us against 81 us at the ml10m shape)
  constexpr int VSC = (V == 4) ? 4 : 2;
  const int ipw = 1;
  const int vs = VSC;
  const int n_sl = (int)hsk_ceil_div(D, 64 * vs);
  // merged launch at a small batch: whole-row item workgroups (n_slices_pad = 0 tells the kernel), see hsk_item_row_body
  static const int rows_on = getenv("HSK_ITEM_ROWS") ? atoi(getenv("HSK_ITEM_ROWS")) : 1;
  // lazy item AdamW (catalogues far beyond the caches): whole rows as well -- the launch is p / m / v traffic on the
  // touched rows, 2 KB contiguous per row and table instead of two 1 KB halves (hbm workload: 1042 -> 977 us), and the
  // slices' L2 affinity buys nothing for a gather that is a tenth of the bytes
  static const int rows_lazy = getenv("HSK_ITEM_ROWS_LAZY") ? atoi(getenv("HSK_ITEM_ROWS_LAZY")) : 1;
  const bool whole_rows = APPLY && ua && rows_on && (n_entries <= 64 * 1024 || (rows_lazy && st->lazy_items)) && !part;
  const int n_slices_pad = whole_rows ? 0 : (n_sl < 8 && 8 % n_sl == 0) ? n_sl : (int)hsk_align_up(n_sl, 8);
  const bool lazy = APPLY && st->lazy_items;
  // lazy item AdamW: only the items with entries (the sort's `touched` list; at most one per entry)
  const int64_t n_list = lazy ? std::min<int64_t>(I, n_entries) : I;
  const unsigned groups = (unsigned)hsk_align_up(hsk_ceil_div(n_list, 4 * ipw), 8);
  const hsk_item_args ia = {Urows, st->item_emb, st->item_bias, st->m_item_emb, st->v_item_emb, st->m_item_bias,
                            st->v_item_bias, urow_index, w.g_s, w.perm, w.offsets, I, K, D, n_slices_pad, ipw, c,
                            gI_out, gIb_out, w.touched, w.n_touched, w.last_step_i, (int)st->step, w.pend,
                            ua ? ua->desc : nullptr, ua ? ua->rel : 0, w.adam_tab, HSK_ADAM_TAB_LEN, n_part};
  const bool gen = APPLY && st->opt_kind != HSK_OPT_ADAMW;   // APPLY == false never calls the update
  const unsigned nblk = groups * (whole_rows ? 1 : n_slices_pad);
  if (APPLY && ua) {
    const int dense = ua->n_users > 0;
    const int nub = (int)hsk_align_up(hsk_ceil_div(dense ? ua->n_users : ua->B, 4) + 1, 8);
    const hsk_ahead_args ahead = aa ? *aa : hsk_ahead_args{};
    // ahead workgroups (4 entries each) in octets, one after every `stride` octets of item workgroups; what does not
    // fit that pattern is dropped (the forward replays those rows itself)
    const int item_oct = (int)(nblk / 8);
    int n_ahead_oct = ahead.coo_user ? (int)hsk_ceil_div(hsk_ceil_div(ahead.n, 4), 8) : 0;
    n_ahead_oct = std::min(n_ahead_oct, item_oct);
    const int stride = n_ahead_oct ? item_oct / n_ahead_oct : 0;
    const int nab_big = ahead.coo_user ? (int)hsk_align_up(hsk_ceil_div(ahead.n, 4), 8) : 0;   // large-batch kernel

#define HSK_ITEM_USER(VS, GEN, LZ)                                                                             \
  do {                                                                                                         \
    if (whole_rows)                                                                                            \
      k_item_user_small<V, NCH, FULL, GEN, LZ><<<nblk + (unsigned)(nub + 8 * n_ahead_oct), 256, 0, stream>>>(    \
          ia, *ua, nub, dense, ahead, n_ahead_oct, stride);                                                    \
    else if (part) {                                                                                           \
      if constexpr (V == 4 && FULL && !LZ)                                                                      \
        k_item_user<V, NCH, FULL, VS, GEN, LZ, true><<<nblk + (unsigned)nub + (unsigned)(ri ? ri->n_total : 0), 256, 0, \
                                                       stream>>>(ia, *ua, nub, dense, ahead, 0,                \
                                                                 ri ? *ri : hsk_ride_item{});                  \
    } else                                                                                                     \
      k_item_user<V, NCH, FULL, VS, GEN, LZ><<<nblk + (unsigned)(nub + (LZ ? nab_big : 0)), 256, 0, stream>>>(   \
          ia, *ua, nub, dense, ahead, LZ ? nab_big : 0);                                                       \
  } while (0)
    if (gen) { if (lazy) HSK_ITEM_USER(VSC, true, true); else HSK_ITEM_USER(VSC, true, false); }
    else     { if (lazy) HSK_ITEM_USER(VSC, false, true); else HSK_ITEM_USER(VSC, false, false); }
#undef HSK_ITEM_USER
    return;
  }
  // lazy item AdamW without the user update in the launch (the sharded step): whole rows, as in the merged launch above
  // (one rank's share of configs[4], 73 % of the 1.25 M x 4 KB rows touched per step: item pass 5.34 -> 4.88 ms, which is
  // what makes the lazy update the faster one there: 7.55-7.63 against 7.92 ms per step for the dense sweep)
  static const int rows_shard = getenv("HSK_ITEM_ROWS_SHARD") ? atoi(getenv("HSK_ITEM_ROWS_SHARD")) : 1;
  if (APPLY && !part && lazy && rows_shard) {
    const unsigned nb = (unsigned)hsk_ceil_div(n_list, 4);
    hsk_item_args ir = ia;
    ir.n_slices_pad = 0;
    if constexpr (APPLY) {
      if (gen) k_item_update_rows<V, NCH, FULL, true, true><<<nb, 256, 0, stream>>>(ir, g_ovf, g_poison);
      else     k_item_update_rows<V, NCH, FULL, false, true><<<nb, 256, 0, stream>>>(ir, g_ovf, g_poison);
    }
    return;
  }
#define HSK_ITEM_SLICED(VS, GEN, LZ)                                                   \
  do {                                                                                 \
    if (part) {                                                                        \
      if constexpr (APPLY && V == 4 && FULL && !LZ)                                    \
        k_item_update_sliced<APPLY, VS, GEN, LZ, true><<<nblk, 256, 0, stream>>>(ia, g_ovf, g_poison);   \
    } else                                                                             \
      k_item_update_sliced<APPLY, VS, GEN, LZ><<<nblk, 256, 0, stream>>>(ia, g_ovf, g_poison);          \
  } while (0)
  if (gen) { if (lazy) HSK_ITEM_SLICED(VSC, true, true); else HSK_ITEM_SLICED(VSC, true, false); }
  else     { if (lazy) HSK_ITEM_SLICED(VSC, false, true); else HSK_ITEM_SLICED(VSC, false, false); }
#undef HSK_ITEM_SLICED
}

// 1 / (number of terms the loss averages over)
static inline double hsk_loss_norm(int kind, double batch, double n_cols) {
  if (kind == HSK_LOSS_BCE) return 1.0 / (batch * n_cols);
  if (kind == HSK_LOSS_SSM) return 1.0 / batch;
  return 1.0 / (batch * (n_cols - 1.0));
}

// device batch construction: positives order[start .. start+batch) + sampled negatives, owner map
static int hsk_launch_prep_sample(const hsk_bprmf_state* st, const hsk_ws& w, const int64_t* order, int64_t start,
                                  int64_t batch, int64_t n_neg, uint64_t stream_id, hipStream_t stream) {
  const hsk_aux* ax = (const hsk_aux*)st->aux;
  const hsk_step_desc* desc = ax ? ax->g_desc : nullptr;
  const int n_part = hsk_part_rule_st(st, batch, n_neg);
  // capture: `stream_id` arrives as the relative step of the batch being prepared
  HSK_STAGE(HSK_STAGE_PREP, k_prep_sample<<<(unsigned)hsk_ceil_div(batch, 4), 256, 0, stream>>>(
                                st->coo_user, st->coo_item, order, start, (int)batch, (int)n_neg, st->csr_indptr,
                                st->csr_indices, (int)st->n_items, st->seed, stream_id, w.u32, w.it32, w.owner, w.cnt,
                                st->status, 0, hsk_alias{st->alias_prob, st->alias_idx}, desc, desc ? (int)stream_id : 0,
                                w.stamp, n_part));
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

static inline bool hsk_capturing(const hsk_bprmf_state* st);

// item sort of the entries in w.it32 -> w.perm / w.offsets
// n_dev: optional device-side entry count (<= total, see hsk_sort_count)
static int hsk_launch_sort(const hsk_bprmf_state* st, const hsk_ws& w, int64_t total, hipStream_t stream,
                           const int* n_dev = nullptr) {
  const int I = (int)st->n_items;
  if (hsk_sort_lds_fits(I, total)) {
    // one workgroup, LDS cursors + per-item fix-up (see k_sort_lds)
    const size_t lds = hsk_sort_lds_bytes(I);
    if (lds > 65536 && !hsk_capturing(st))   // opt-in for > 64 KB of dynamic LDS (not a stream operation)
      HSK_HIP(hipFuncSetAttribute((const void*)k_sort_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HSK_STAGE(HSK_STAGE_SCATTER, (k_sort_lds<<<1, 1024, lds, stream>>>(w.it32, (int)total, I, w.perm, w.offsets,
                                                                       st->lazy_items ? w.touched : nullptr,
                                                                       st->lazy_items ? w.n_touched : nullptr, n_dev)));
    HSK_LAUNCH_CHECK();
    return HSK_OK;
  }
  if (total <= 1024 * 8) {
    // one workgroup sorts the whole batch (see k_sort_small)
    int nbits = 1;
    while ((1ll << nbits) <= (long long)I) ++nbits;   // keys 0..I (I = padding) fit
    int* tl = st->lazy_items ? w.touched : nullptr;
    int* tn = st->lazy_items ? w.n_touched : nullptr;
    HSK_STAGE(HSK_STAGE_SCATTER, {
      if (total <= 1024 * 2)
        k_sort_small<2><<<1, 1024, 0, stream>>>(w.it32, (int)total, I, nbits, w.perm, w.offsets, tl, tn, n_dev);
      else if (total <= 1024 * 4)
        k_sort_small<4><<<1, 1024, 0, stream>>>(w.it32, (int)total, I, nbits, w.perm, w.offsets, tl, tn, n_dev);
      else
        k_sort_small<8><<<1, 1024, 0, stream>>>(w.it32, (int)total, I, nbits, w.perm, w.offsets, tl, tn, n_dev);
    });
    HSK_LAUNCH_CHECK();
    return HSK_OK;
  }
  hsk_sort_plan plan;
  HSK_REQUIRE(hsk_make_sort_plan(I, total, &plan) == 0, HSK_ERR_UNSUPPORTED, "item sort: n_items too large");
  const size_t bucket_lds = ((size_t)(st->lazy_items ? 6 : 5) * plan.ipb + 2) * sizeof(int);
  if (bucket_lds > 65536)
    HSK_HIP(hipFuncSetAttribute((const void*)k_sort_bucket, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bucket_lds));
  if (st->lazy_items) HSK_HIP(hipMemsetAsync(w.n_touched, 0, sizeof(int), stream));
  HSK_STAGE(HSK_STAGE_SCAN, {
    k_sort_hist<<<(unsigned)hsk_ceil_div(plan.n_units, 4), 256, 0, stream>>>(w.it32, (int)total, plan, w.hist, n_dev);
    k_sort_rowscan<<<(unsigned)hsk_ceil_div(plan.n_buckets, 4), 256, 0, stream>>>(w.hist, plan, w.btot);
  });
  HSK_LAUNCH_CHECK();
  HSK_STAGE(HSK_STAGE_SCATTER, {
    k_sort_scatter<<<(unsigned)hsk_ceil_div(plan.n_units, 4), 256, 0, stream>>>(w.it32, (int)total, plan, w.hist, w.btot,
                                                                                 w.perm1, w.bstart, n_dev);
    k_sort_bucket<<<(unsigned)plan.n_buckets, 256, bucket_lds, stream>>>(
        w.perm1, (int)total, I, plan, w.bstart, w.perm, w.offsets, st->lazy_items ? w.touched : nullptr,
        st->lazy_items ? w.n_touched : nullptr, n_dev);
  });
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

static inline bool hsk_capturing(const hsk_bprmf_state* st) {
  const hsk_aux* a = (const hsk_aux*)st->aux;
  return a && a->g_desc;
}

// a prefetched batch that the next call does not consume: give its owner map back
static int hsk_discard_prefetch(hsk_bprmf_state* st, const hsk_ws& w_all, hipStream_t stream) {
  hsk_aux* a = (hsk_aux*)st->aux;
  if (!a || !a->pf_valid) return HSK_OK;
  const hsk_ws w = hsk_select(w_all, a->cur_set ^ 1);
  HSK_HIP(hipStreamWaitEvent(stream, a->ev_ready, 0));
  k_release_owner<<<(unsigned)hsk_ceil_div(a->pf_batch, 256), 256, 0, stream>>>(w.u32, (int)a->pf_batch, w.owner, w.cnt);
  HSK_LAUNCH_CHECK();
  a->pf_valid = false;
  return HSK_OK;
}

// the next batch, if the caller named it: sampled and sorted on the side stream from here on
// Fork point.  A forward kernel of >= 512 workgroups occupies most wave slots of the chip for its whole duration:
// side-stream kernels forked before it only get in as it drains, and delay its workgroups (measured at B=4096,
// D=512: 243 us/step forked before, 231 us forked after, 252 us without prefetch).  A small forward kernel leaves
// the chip mostly idle and the step is launch-latency-bound: fork as early as possible (B=128: 65 us before, 97 us
// after, 83 us without).
static bool hsk_pf_early(int64_t B) {
  static const int env = getenv("HSK_PF_EARLY") ? atoi(getenv("HSK_PF_EARLY")) : -1;   // experiments: 0 / 1 force
  return env >= 0 ? env != 0 : B < 2048;
}
#define HSK_PREFETCH_MIN_ENTRIES 4096

// true: this step will fork a prefetch (hint present and worth it)
static bool hsk_prefetch_wanted(const hsk_bprmf_state* st) {
  hsk_aux* aux = (hsk_aux*)st->aux;
  if (!aux || !aux->hint_valid) return false;
  if (aux->pipe_active) return false;   // the preparation rides in the step's own launches (hsk_pipe_step)
  if (aux->grouped) return false; // the run's batches are prepared G at a time by the capture loop (hsk_capture_steps)
  if (aux->g_desc) return true;   // inside a graph the fork / join are dependencies, not host calls
  if (aux->hint_batch * (aux->hint_nneg + 1) < HSK_PREFETCH_MIN_ENTRIES) {
    aux->hint_valid = false;  // a few hundred entries: the fork/join events cost more than the five tiny kernels
    return false;
  }
  return true;
}

// fork_recorded: ev_fork already rides on the forward kernel's completion signal (hipExtLaunchKernelGGL stop event),
// which saves the separate barrier packet a hipEventRecord would put between the forward and the item pass
// fork_on: the event to fork on when it is not aux->ev_fork (a timed step: the stop event of the timed forward launch)
static int hsk_launch_prefetch(hsk_bprmf_state* st, const hsk_ws& w_all, int set, hipStream_t stream,
                               bool fork_recorded = false, hipEvent_t fork_on = nullptr) {
  hsk_aux* aux = (hsk_aux*)st->aux;
  if (!hsk_prefetch_wanted(st)) return HSK_OK;
  const hsk_ws wn = hsk_select(w_all, set ^ 1);
  if (!fork_recorded) HSK_HIP(hipEventRecord(aux->ev_fork, stream));
  HSK_HIP(hipStreamWaitEvent(aux->side, fork_on ? fork_on : aux->ev_fork, 0));
  int prc = hsk_launch_prep_sample(st, wn, aux->hint_order, aux->hint_start, aux->hint_batch, aux->hint_nneg,
                                   aux->g_desc ? (uint64_t)(aux->g_rel + 1) : (uint64_t)st->step, aux->side);
  if (!prc) {
    const int64_t tot = aux->hint_batch * hsk_part_cols(aux->hint_nneg + 1, hsk_part_rule_st(st, aux->hint_batch, aux->hint_nneg));
    prc = hsk_launch_sort(st, wn, tot, aux->side);
  }
  if (prc) return prc;
  {   // experiment (HSK_SIDE_DUMMY=n): n empty launches behind the sort -- what a kernel boundary on the side stream costs
    static const int n_dummy = getenv("HSK_SIDE_DUMMY") ? atoi(getenv("HSK_SIDE_DUMMY")) : 0;
    for (int i = 0; i < n_dummy && !aux->g_desc; ++i) k_fill_i32<<<1, 64, 0, aux->side>>>(nullptr, 0, 0);
  }
  HSK_HIP(hipEventRecord(aux->ev_ready, aux->side));
  aux->pf_valid = true;
  aux->pf_order = aux->hint_order;
  aux->pf_start = aux->hint_start;
  aux->pf_batch = aux->hint_batch;
  aux->pf_nneg = aux->hint_nneg;
  aux->pf_step = st->step;
  aux->hint_valid = false;
  return HSK_OK;
}

// stages after prep filled u32 / it32 / owner / cnt of `w` (and, with sorted == true, the item sort too)
// n_part > 1: batch rows in the partitioned layout (the positive in n_part columns) -> item-partitioned forward
// slot: the batch's slot inside the set (grouped preparation)
static int hsk_run_step(hsk_bprmf_state* st, const hsk_ws& w_all, int set, bool sorted, int64_t B, int64_t K,
                        hipStream_t stream, int n_part = 1, int slot = 0) {
  const hsk_ws w = hsk_select(w_all, set, slot);
  const int64_t K_real = K;
  K = hsk_part_cols(K_real, n_part);   // columns of the batch rows: the positive n_part times (k_prep_sample)
  const int64_t total = B * K;
  const int U = (int)st->n_users, D = (int)st->dim;
  st->step += 1;
  const hsk_adamw_consts c = hsk_make_adamw_consts(st->lr, st->beta1, st->beta2, st->eps, st->wd, st->step, st->opt_kind);
  const double inv_bn_d = hsk_loss_norm(st->loss_kind, (double)B, (double)K_real);
  const float inv_bn = (float)inv_bn_d;
  hsk_aux* aux = (hsk_aux*)st->aux;
  const bool gen = st->opt_kind != HSK_OPT_ADAMW;   // generic optimiser arithmetic instead of the AdamW-only kernels
  if (aux) {
    aux->cur_set = aux->last_set = set;
    aux->cur_slot = aux->last_slot = slot;
  }
  if (!sorted) {
    int src = hsk_launch_sort(st, w, total, stream);
    if (src) return src;
  }
  if (st->lazy_items) {
    // the forward must read current item rows too: replay the missed zero-gradient steps of this batch's items
    // (the list the sort left in w.touched)
    const unsigned nblk = (unsigned)std::min<int64_t>(st->n_items, total);
    if (gen)
      HSK_STAGE(HSK_STAGE_ITEM, (k_item_catch_up<2, true><<<nblk, 256, 0, stream>>>(
                                    st->item_emb, st->m_item_emb, st->v_item_emb, st->item_bias, st->m_item_bias,
                                    st->v_item_bias, w.touched, w.n_touched, w.last_step_i, D, (int)st->step, c,
                                    w.adam_tab, HSK_ADAM_TAB_LEN, aux ? aux->g_desc : nullptr, aux ? aux->g_rel : 0, w.pend)));
    else
      HSK_STAGE(HSK_STAGE_ITEM, (k_item_catch_up<2, false><<<nblk, 256, 0, stream>>>(
                                    st->item_emb, st->m_item_emb, st->v_item_emb, st->item_bias, st->m_item_bias,
                                    st->v_item_bias, w.touched, w.n_touched, w.last_step_i, D, (int)st->step, c,
                                    w.adam_tab, HSK_ADAM_TAB_LEN, aux ? aux->g_desc : nullptr, aux ? aux->g_rel : 0, w.pend)));
    HSK_LAUNCH_CHECK();
  }
  if (hsk_pf_early(B)) {
    int prc = hsk_launch_prefetch(st, w_all, set, stream);
    if (prc) return prc;
  }
  // late fork: the fork event is the forward kernel's own completion signal (an ordinary event record in a capture)
  const bool capturing = aux && aux->g_desc;
  const hsk_step_desc* gdesc = capturing ? aux->g_desc : nullptr;
  const int grel = capturing ? aux->g_rel : 0;
  const bool late_fork = !hsk_pf_early(B) && hsk_prefetch_wanted(st);
  hipEvent_t fork_ev = (late_fork && !capturing) ? aux->ev_fork : nullptr;

  hipEvent_t timed_fwd_end = nullptr;   // a timed step: the forward launch's stop event doubles as the fork event
  int rc = hsk_dispatch_dim(D, [&](auto v_, auto n_, auto f_) {
    constexpr int V = decltype(v_)::value;
    constexpr int NCH = decltype(n_)::value;
    constexpr bool FULL = decltype(f_)::value;
    constexpr int R = (V * NCH >= 16) ? 2 : HSK_FWD_R;
    // lazy user AdamW: the forward replays the pending zero-gradient steps of its user row in registers and leaves
    // the current row in ucur (no separate catch-up launch, no rewrite of the row before the owner's update).
    // st->catchup_apart: a stand-alone catch-up launch in front instead -- at B = 4096 the replay is ~9 us of pure VALU
    // work at the head of every forward wave (forward 103 -> 112 us) but the extra launch costs more than it saves
    // (224.5 vs 219.3 us per step), so it is off by default; with it the forward is the pure gather.
    const bool apart = st->lazy_users && !capturing && st->catchup_apart != 0;
    if (apart) {
#define HSK_CATCH_UP(VV, GEN)                                                                                     \
  HSK_STAGE(HSK_STAGE_USER, (k_user_catch_up<VV, GEN><<<(unsigned)B, 256, 0, stream>>>(                            \
                                st->user_emb, st->m_user_emb, st->v_user_emb, st->user_bias, st->m_user_bias,      \
                                st->v_user_bias, w.u32, w.owner, w.last_step, (int)B, D, (int)st->step, c,         \
                                w.adam_tab, HSK_ADAM_TAB_LEN, nullptr, nullptr)))
      if (D % 2 == 0) {
        if (gen) HSK_CATCH_UP(2, true); else HSK_CATCH_UP(2, false);
      } else {
        if (gen) HSK_CATCH_UP(1, true); else HSK_CATCH_UP(1, false);
      }
#undef HSK_CATCH_UP
    }
    // The forward kernel is launched through hipExtLaunchKernelGGL, whose start / stop events are the dispatch's own
    // timestamps: on a timed step they ARE the stage timing (kernel time as rocprofv3 reports it, no barrier packets
    // around the launch); otherwise the stop event is the prefetch's fork event.
    hsk_lazy_user_args lz = {};
    if (st->lazy_users)
      lz = hsk_lazy_user_args{st->m_user_emb, st->v_user_emb, w.last_step, w.owner, w.dupcnt, w.duplist, w.ucur,
                              w.mcur, w.vcur, (int)st->step, gen ? 1 : 0, c, w.adam_tab, HSK_ADAM_TAB_LEN, gdesc, grel};
    else if (D % 2 == 0)
      lz.ucur = w.ucur;   // dense users in the item pass's launch: it reads the batch's rows from ucur
    hipEvent_t fwd_beg = nullptr, fwd_end = fork_ev;
    hsk_timing* tm = (hsk_timing*)st->timing;
    const bool time_fwd = tm && st->timing_now && ((st->timing_mask >> HSK_STAGE_FWD) & 1);
    if (time_fwd) {
      fwd_beg = tm->get();
      fwd_end = tm->get();
      if (!fwd_beg || !fwd_end) return HSK_ERR_HIP;
      tm->beg[HSK_STAGE_FWD].push_back(fwd_beg);
      tm->end[HSK_STAGE_FWD].push_back(fwd_end);
      timed_fwd_end = fwd_end;
    }
#define HSK_LAUNCH_FWD(LK)                                                                                          \
  if (capturing)                                                                                                    \
    hipLaunchKernelGGL((k_fwd_ugrad<V, NCH, FULL, R, LK>), dim3((unsigned)hsk_ceil_div(B, 4)), dim3(256), 0, stream,   \
                       (const float*)st->user_emb, (const float*)st->item_emb, (const float*)st->item_bias,         \
                       (const int*)w.u32, (const int*)w.it32, (int)B, (int)K, D, inv_bn, (float)st->ssm_log_adjust,  \
                       w.g_s, w.dUb, w.loss_b, (const int*)nullptr, lz);                                            \
  else                                                                                                              \
    hipExtLaunchKernelGGL((k_fwd_ugrad<V, NCH, FULL, R, LK>), dim3((unsigned)hsk_ceil_div(B, 4)), dim3(256), 0, stream, \
                          fwd_beg, fwd_end, 0, (const float*)st->user_emb, (const float*)st->item_emb,              \
                          (const float*)st->item_bias, (const int*)w.u32, (const int*)w.it32, (int)B, (int)K, D,    \
                          inv_bn, (float)st->ssm_log_adjust, w.g_s, w.dUb, w.loss_b, (const int*)nullptr, lz)
    if (n_part > 1) {
      // item-partitioned forward; the NEXT batch's user rows are brought up to date by leading workgroups of the same
      // launch (pure VALU work beside a kernel that waits on the L2 / the fabric), see hsk_fwd_part.h
      if constexpr (V == 4 && FULL) {
        constexpr int RP = (V * NCH >= 16) ? 2 : HSK_FWD_PART_R;
        hsk_ahead_args aa = {};
        int n_ahead = 0;
        static const int ahead_on = getenv("HSK_AHEAD") ? atoi(getenv("HSK_AHEAD")) : 1;
        if (st->lazy_users && aux && ahead_on && !apart && (aux->pf_valid || aux->hint_valid)) {
          const bool pf = aux->pf_valid;
          aa = hsk_ahead_args{st->coo_user, pf ? aux->pf_order : aux->hint_order, pf ? aux->pf_start : aux->hint_start,
                              (int)(pf ? aux->pf_batch : aux->hint_batch), w.stamp, w.claim, st->user_emb, st->m_user_emb,
                              st->v_user_emb, st->user_bias, st->m_user_bias, st->v_user_bias, w.last_step, D,
                              (int)st->step, c, w.adam_tab, HSK_ADAM_TAB_LEN, gdesc, grel};
          n_ahead = (int)hsk_align_up(hsk_ceil_div(aa.n, 4), 8);
        }
        const unsigned n_unit = 8u * (unsigned)hsk_ceil_div(hsk_ceil_div(B, 4), 8 / n_part);
        // HSK_AHEAD_MIX=1: the ahead octets interleaved with the unit octets instead of leading (measured: 88 vs 86 us)
        static const int ahead_mix = getenv("HSK_AHEAD_MIX") ? atoi(getenv("HSK_AHEAD_MIX")) : 0;
        const int stride = (ahead_mix && n_ahead > 0) ? (int)(n_unit / 8) / (n_ahead / 8) : 0;
        const hsk_part_args pa = {n_part, (int)st->n_items, n_ahead, stride};
        const hsk_ride_fwd rf = (aux && aux->ride_fwd) ? *aux->ride_fwd : hsk_ride_fwd{};   // riding preparation phases
        const unsigned grid = (unsigned)rf.n_total + (unsigned)n_ahead + n_unit;
#define HSK_LAUNCH_FWD_PART(LK, GEN)                                                                                 \
  if (capturing)                                                                                                    \
    hipLaunchKernelGGL((k_fwd_part<V, NCH, FULL, RP, LK, GEN>), dim3(grid), dim3(256), 0, stream,                    \
                       (const float*)st->user_emb, (const float*)st->item_emb, (const float*)st->item_bias,         \
                       (const int*)w.u32, (const int*)w.it32, (int)B, (int)K, D, inv_bn, w.g_s, w.dUb, w.loss_b, lz,  \
                       pa, aa, rf);                                                                                 \
  else                                                                                                              \
    hipExtLaunchKernelGGL((k_fwd_part<V, NCH, FULL, RP, LK, GEN>), dim3(grid), dim3(256), 0, stream, fwd_beg, fwd_end, \
                          0, (const float*)st->user_emb, (const float*)st->item_emb, (const float*)st->item_bias,   \
                          (const int*)w.u32, (const int*)w.it32, (int)B, (int)K, D, inv_bn, w.g_s, w.dUb, w.loss_b,  \
                          lz, pa, aa, rf)
        if (st->loss_kind == HSK_LOSS_BCE) {
          if (gen) { HSK_LAUNCH_FWD_PART(HSK_LOSS_BCE, true); } else { HSK_LAUNCH_FWD_PART(HSK_LOSS_BCE, false); }
        } else {
          if (gen) { HSK_LAUNCH_FWD_PART(HSK_LOSS_BPR, true); } else { HSK_LAUNCH_FWD_PART(HSK_LOSS_BPR, false); }
        }
#undef HSK_LAUNCH_FWD_PART
        return HSK_OK;
      } else {
        hsk_set_error("internal: item-partitioned forward selected for dim %d", D);
        return HSK_ERR_UNSUPPORTED;
      }
    }
    // small batches: a workgroup per positive (hsk_fwd_small.h); the one-wave kernel would leave most SIMDs idle
    // and walk each positive's rows as a chain of memory latencies
#define HSK_LAUNCH_FWD_WG(LK, NW)                                                                                   \
  if (capturing)                                                                                                    \
    hipLaunchKernelGGL((k_fwd_ugrad_wg<V, NCH, FULL, R, LK, NW>), dim3((unsigned)B), dim3(64 * NW),                    \
                       (unsigned)(NW * (size_t)D * sizeof(float)), stream, (const float*)st->user_emb,              \
                       (const float*)st->item_emb, (const float*)st->item_bias, (const int*)w.u32,                  \
                       (const int*)w.it32, (int)B, (int)K, D, inv_bn, w.g_s, w.dUb, w.loss_b, lz);                   \
  else                                                                                                              \
    hipExtLaunchKernelGGL((k_fwd_ugrad_wg<V, NCH, FULL, R, LK, NW>), dim3((unsigned)B), dim3(64 * NW),                 \
                          (unsigned)(NW * (size_t)D * sizeof(float)), stream, fwd_beg, fwd_end, 0,                   \
                          (const float*)st->user_emb, (const float*)st->item_emb, (const float*)st->item_bias,      \
                          (const int*)w.u32, (const int*)w.it32, (int)B, (int)K, D, inv_bn, w.g_s, w.dUb, w.loss_b, lz)
    // (also for very few columns: idle waves of the workgroup cost nothing at B = 128 and the kernel's loads come in two
    // rounds instead of four -- ml100k shape, K = 2: 13.8 -> 12.2 us per step; HSK_WG_MINK=9 restores the old rule)
    static const int wg_min_k = getenv("HSK_WG_MINK") ? atoi(getenv("HSK_WG_MINK")) : 2;
    const bool wg_fwd = B <= 1024 && st->loss_kind != HSK_LOSS_SSM && K >= wg_min_k;
    const bool wg8 = wg_fwd && K >= 33 && B <= 256;   // few, wide positives: 8 waves each
    if (wg_fwd && st->loss_kind == HSK_LOSS_BCE) {
      if (wg8) { HSK_LAUNCH_FWD_WG(HSK_LOSS_BCE, 8); } else { HSK_LAUNCH_FWD_WG(HSK_LOSS_BCE, 4); }
    } else if (wg_fwd) {
      if (wg8) { HSK_LAUNCH_FWD_WG(HSK_LOSS_BPR, 8); } else { HSK_LAUNCH_FWD_WG(HSK_LOSS_BPR, 4); }
    } else if (st->loss_kind == HSK_LOSS_BCE) {
      HSK_LAUNCH_FWD(HSK_LOSS_BCE);
    } else if (st->loss_kind == HSK_LOSS_SSM) {
      HSK_LAUNCH_FWD(HSK_LOSS_SSM);
    } else {
      HSK_LAUNCH_FWD(HSK_LOSS_BPR);
    }
#undef HSK_LAUNCH_FWD
#undef HSK_LAUNCH_FWD_WG
    return HSK_OK;
  });
  if (rc) return rc;
  HSK_LAUNCH_CHECK();

  if (late_fork) {
    // the fork rides on the forward launch's own stop event: aux->ev_fork, or (timed step) the timing event
    const bool fork_on_kernel = !capturing;
    int prc = hsk_launch_prefetch(st, w_all, set, stream, fork_on_kernel, timed_fwd_end);
    if (prc) return prc;
  }

  rc = hsk_dispatch_dim(D, [&](auto v_, auto n_, auto f_) {
    constexpr int V = decltype(v_)::value;
    constexpr int NCH = decltype(n_)::value;
    constexpr bool FULL = decltype(f_)::value;
    constexpr int R = (V * NCH >= 16) ? 2 : HSK_FWD_R;
    // user update: the owners' rows (lazy: pending steps were replayed by the forward) or a dense sweep over the
    // table; + one workgroup for the loss reduction / global bias.  Even D: in the item pass's own launch.
    const hsk_finish_args fin = {w.loss_b, (int)(B * n_part), inv_bn_d, st->loss_out, st->global_bias,
                                 st->m_global_bias, st->v_global_bias};   // partitioned forward: [n_part, B] partial terms
    const bool lazy = st->lazy_users != 0;
    const hsk_user_lazy_args ua = {st->user_emb, st->m_user_emb, st->v_user_emb, st->user_bias, st->m_user_bias,
                                   st->v_user_bias, w.dUb, w.u32, w.owner, w.cnt, w.last_step, (int)B, D,
                                   (int)st->step, c, fin, w.dupcnt, w.duplist, lazy ? w.adam_tab : nullptr,
                                   HSK_ADAM_TAB_LEN, w.ucur, w.mcur, w.vcur, lazy ? 0 : U, gdesc, grel, w.adam_tab,
                                   HSK_ADAM_TAB_LEN, n_part};
    const bool timed_apart = st->timing && st->timing_now &&
                             (((st->timing_mask >> HSK_STAGE_ITEM) | (st->timing_mask >> HSK_STAGE_USER)) & 1);
    if (D % 2 == 0 && !timed_apart) {
      // lazy users, small batches: the NEXT batch (the prefetch's, or the pending hint's) gets its rows brought up to
      // date by workgroups of this same launch (hsk_user_ahead_body).  A small step is a chain of latencies with the
      // VALUs idle, and the replay leaves the head of the forward's critical path (ml1m shape: forward 20 -> 11 us).
      // At B = 4096 every kernel is within reach of its issue limit and the replay costs its ~10 us of VALU time
      // wherever it runs: measured 233 us per step with these workgroups in the item pass against 221 us with the
      // replay inside the forward, so large batches keep it there.
      hsk_ahead_args aa = {};
      static const int ahead_on = getenv("HSK_AHEAD") ? atoi(getenv("HSK_AHEAD")) : 1;
      // (huge catalogues with lazy item AdamW: the item launch is AdamW traffic on the touched rows, memory-bound with
      // idle VALUs, so the replay hides there as well)
      if (lazy && aux && ahead_on && (hsk_pf_early(B) || st->lazy_items) && (aux->pf_valid || aux->hint_valid)) {
        const bool pf = aux->pf_valid;
        aa = hsk_ahead_args{st->coo_user, pf ? aux->pf_order : aux->hint_order, pf ? aux->pf_start : aux->hint_start,
                            (int)(pf ? aux->pf_batch : aux->hint_batch), w.stamp, w.claim, st->user_emb, st->m_user_emb,
                            st->v_user_emb, st->user_bias, st->m_user_bias, st->v_user_bias, w.last_step, D,
                            (int)st->step, c, w.adam_tab, HSK_ADAM_TAB_LEN, gdesc, grel};
      }
      // the item pass reads the user rows from ucur, the user blocks rewrite the table: independent -> one launch
      hsk_launch_item_pass<V, NCH, FULL, R, true>(st, w, w.ucur, nullptr, (int)K, c, nullptr, nullptr, stream, total, &ua,
                                                  &aa, n_part, nullptr, nullptr, aux ? aux->ride_item : nullptr);
    } else {
      const bool from_ucur = lazy || D % 2 == 0;
      HSK_STAGE(HSK_STAGE_ITEM, (hsk_launch_item_pass<V, NCH, FULL, R, true>(
                                    st, w, from_ucur ? w.ucur : st->user_emb, from_ucur ? nullptr : w.u32, (int)K, c,
                                    nullptr, nullptr, stream, total, nullptr, nullptr, n_part)));
      if (n_part > 1) {   // partial gradient rows: the PART flavours (V == 4 && FULL by the partition rule)
        if constexpr (V == 4 && FULL) {
          const unsigned gl = (unsigned)hsk_ceil_div(B, 4) + 1, gd = (unsigned)hsk_ceil_div(U, 4) + 1;
          if (lazy) {
            if (gen) HSK_STAGE(HSK_STAGE_USER, (k_user_update_lazy<V, NCH, FULL, true, true><<<gl, 256, 0, stream>>>(ua)));
            else     HSK_STAGE(HSK_STAGE_USER, (k_user_update_lazy<V, NCH, FULL, false, true><<<gl, 256, 0, stream>>>(ua)));
          } else {
            if (gen) HSK_STAGE(HSK_STAGE_USER, (k_user_update_dense<V, NCH, FULL, true, true><<<gd, 256, 0, stream>>>(ua)));
            else     HSK_STAGE(HSK_STAGE_USER, (k_user_update_dense<V, NCH, FULL, false, true><<<gd, 256, 0, stream>>>(ua)));
          }
        }
      } else if (lazy) {
        if (gen)
          HSK_STAGE(HSK_STAGE_USER, (k_user_update_lazy<V, NCH, FULL, true><<<(unsigned)hsk_ceil_div(B, 4) + 1, 256, 0, stream>>>(ua)));
        else
          HSK_STAGE(HSK_STAGE_USER, (k_user_update_lazy<V, NCH, FULL, false><<<(unsigned)hsk_ceil_div(B, 4) + 1, 256, 0, stream>>>(ua)));
      } else {
        if (gen)
          HSK_STAGE(HSK_STAGE_USER, (k_user_update_dense<V, NCH, FULL, true><<<(unsigned)hsk_ceil_div(U, 4) + 1, 256, 0, stream>>>(ua)));
        else
          HSK_STAGE(HSK_STAGE_USER, (k_user_update_dense<V, NCH, FULL, false><<<(unsigned)hsk_ceil_div(U, 4) + 1, 256, 0, stream>>>(ua)));
      }
    }
    return HSK_OK;
  });
  if (rc) return rc;
  HSK_LAUNCH_CHECK();
  if (!capturing && (st->lazy_users || st->lazy_items)) {
    int frc = hsk_periodic_flush(st, w, stream, st->step - 1, (double)B, (double)(B * K_real));
    if (frc) return frc;
  }
  return HSK_OK;
}

static int hsk_check_batch(const hsk_bprmf_state* st, int64_t batch, int64_t n_cols) {
  HSK_REQUIRE(!st->ws_sharded, HSK_ERR_INVALID, "state carved for the sharded step (ws_sharded = 1): use hsk_shard_*");
  HSK_REQUIRE(batch > 0 && batch <= st->max_batch, HSK_ERR_INVALID, "batch %lld outside (0, %lld]", (long long)batch,
              (long long)st->max_batch);
  HSK_REQUIRE(n_cols >= 2 && n_cols <= st->max_cols, HSK_ERR_INVALID, "n_cols %lld outside [2, %lld]",
              (long long)n_cols, (long long)st->max_cols);
  return HSK_OK;
}

// =============================================================================================
// in-launch preparation pipeline (large batches on the item-partitioned forward)
// =============================================================================================
// The side-stream prefetch above costs the step more than the work it overlaps: timelines of the ml10m step
// (profiles/r4_ml10m) show ~5 us between the forward and the item/user launch (the fork event riding on the forward's
// completion signal), 8-12 us between the item/user launch and the next forward (the join on the side stream's event),
// and the item/user launch stretched from 70 to 95 us by five high-priority launches trickling through it -- each empty
// launch added to that stream costs the step another 2.5 us.  Here the five preparation phases of a batch
//   S sample   H per-unit bucket histogram   R per-bucket scan over the units   C scatter by bucket   B sort inside buckets
// run as EXTRA WORKGROUPS AT THE HEAD of the two launches a step has anyway (hsk_ride_fwd / hsk_ride_item), two steps
// ahead: the forward of step t carries B of batch t, S of batch t+2 and R of batch t+1; its item/user launch C of batch
// t+1 and H of batch t+2.  The launch boundaries between them are the phases' barriers; there is no side stream, no
// event and no launch of its own on the step's path.  Three buffer sets rotate.  Same kernels' bodies, same draws, same
// sort: bit-identical batches, losses and parameters (tests/test_hip_parity.py).
// What is not prepared when its step comes (the first steps of a run, a batch the caller did not name ahead of time) is
// brought up to date by stand-alone launches of the same phases on the step's own stream.
struct hsk_batch_desc {
  bool valid;
  const int64_t* order;
  int64_t start, batch, n_neg, step;
};

static bool hsk_pipe_plan(const hsk_bprmf_state* st, int64_t batch, int64_t n_neg, int* n_part_out, hsk_sort_plan* plan,
                          int64_t* total_out) {
  const int n_part = hsk_part_rule_st(st, batch, n_neg);
  const int64_t total = batch * hsk_part_cols(n_neg + 1, n_part);
  if (n_part_out) *n_part_out = n_part;
  if (total_out) *total_out = total;
  if (n_part <= 1 || hsk_sort_lds_fits(st->n_items, total) || total <= 1024 * 8) return false;
  hsk_sort_plan p;
  if (hsk_make_sort_plan(st->n_items, total, &p) != 0 || p.ipb > HSK_PIPE_MAX_IPB) return false;
  if (plan) *plan = p;
  return true;
}

// HSK_PIPE=0 / hsk_bprmf_set_pipeline(0): the side-stream prefetch everywhere (A/B runs, and the reference the
// bit-equality tests hold the pipeline to)
static int hsk_pipe_switch = getenv("HSK_PIPE") ? atoi(getenv("HSK_PIPE")) : 1;
extern "C" void hsk_bprmf_set_pipeline(int on) { hsk_pipe_switch = on ? 1 : 0; }

static bool hsk_pipe_eligible(const hsk_bprmf_state* st, int64_t batch, int64_t n_neg) {
  const hsk_aux* a = (const hsk_aux*)st->aux;
  if (!hsk_pipe_switch || !a || a->g_desc || st->ws_sharded || st->lazy_items) return false;
  if (st->loss_kind != HSK_LOSS_BPR && st->loss_kind != HSK_LOSS_BCE) return false;
  if (st->timing && (st->timing_mask & ~(1 << HSK_STAGE_FWD))) return false;   // stage timing brackets the separate launches
  return hsk_pipe_plan(st, batch, n_neg, nullptr, nullptr, nullptr);
}

// one phase of the preparation of the batch in `b`, as a launch of its own
static int hsk_pipe_launch_phase(const hsk_bprmf_state* st, const hsk_ws& w, const hsk_aux::pipe_slot& b, int phase,
                                 hipStream_t stream) {
  hsk_sort_plan plan;
  int64_t total = 0;
  HSK_REQUIRE(hsk_pipe_plan(st, b.batch, b.n_neg, nullptr, &plan, &total), HSK_ERR_UNSUPPORTED, "internal: pipeline shape");
  const int I = (int)st->n_items;
  switch (phase) {
    case HSK_PIPE_S:
      return hsk_launch_prep_sample(st, w, b.order, b.start, b.batch, b.n_neg, (uint64_t)b.step, stream);
    case HSK_PIPE_H:
      k_sort_hist<<<(unsigned)hsk_ceil_div(plan.n_units, 4), 256, 0, stream>>>(w.it32, (int)total, plan, w.hist, nullptr);
      break;
    case HSK_PIPE_R:
      k_sort_rowscan<<<(unsigned)hsk_ceil_div(plan.n_buckets, 4), 256, 0, stream>>>(w.hist, plan, w.btot);
      break;
    case HSK_PIPE_C:
      k_sort_scatter<<<(unsigned)hsk_ceil_div(plan.n_units, 4), 256, 0, stream>>>(w.it32, (int)total, plan, w.hist, w.btot,
                                                                                   w.perm1, w.bstart, nullptr);
      break;
    default:
      k_sort_bucket<<<(unsigned)plan.n_buckets, 256, (5 * (size_t)plan.ipb + 2) * sizeof(int), stream>>>(
          w.perm1, (int)total, I, plan, w.bstart, w.perm, w.offsets, nullptr, nullptr, nullptr);
  }
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

static hsk_ride_sort hsk_pipe_ride_sort(const hsk_bprmf_state* st, const hsk_ws& w, const hsk_aux::pipe_slot& b, int phase) {
  hsk_sort_plan plan;
  int64_t total = 0;
  (void)hsk_pipe_plan(st, b.batch, b.n_neg, nullptr, &plan, &total);
  const int n_blocks = phase == HSK_PIPE_B ? plan.n_buckets
                                           : (int)hsk_ceil_div(phase == HSK_PIPE_R ? plan.n_buckets : plan.n_units, 4);
  return hsk_ride_sort{n_blocks, w.it32, (int)total, (int)st->n_items, plan, w.hist, w.btot, w.perm1, w.bstart, w.perm, w.offsets};
}

// give the owner map of a sampled batch that will not be trained on back, forget the slot
static int hsk_pipe_drop(hsk_bprmf_state* st, const hsk_ws& w_all, int slot, hipStream_t stream) {
  hsk_aux* a = (hsk_aux*)st->aux;
  hsk_aux::pipe_slot& b = a->pipe[slot];
  if (b.valid && b.stage >= HSK_PIPE_S) {
    const hsk_ws w = hsk_select(w_all, slot);
    k_release_owner<<<(unsigned)hsk_ceil_div(b.batch, 256), 256, 0, stream>>>(w.u32, (int)b.batch, w.owner, w.cnt);
    HSK_LAUNCH_CHECK();
  }
  b = hsk_aux::pipe_slot{};
  return HSK_OK;
}

// leaves the pipeline empty (a flush, a step issued through another path, a re-initialised workspace)
static int hsk_pipe_reset(hsk_bprmf_state* st, const hsk_ws& w_all, hipStream_t stream) {
  hsk_aux* a = (hsk_aux*)st->aux;
  if (!a) return HSK_OK;
  for (int i = 0; i < 3; ++i) {
    int rc = hsk_pipe_drop(st, w_all, i, stream);
    if (rc) return rc;
  }
  if (a->cur_set > 1) a->cur_set = 0;   // (the other paths alternate between sets 0 and 1; the sets are scratch)
  return HSK_OK;
}

static bool hsk_pipe_same(const hsk_aux::pipe_slot& b, const hsk_batch_desc& d) {
  return b.valid && d.valid && b.order == d.order && b.start == d.start && b.batch == d.batch && b.n_neg == d.n_neg &&
         b.step == d.step;
}

// ONE step of a pipelined run: `cur` is trained on now (cur.step == st->step); n1 / n2 are the batches of the next two
// steps where the caller knows them.
static int hsk_pipe_step(hsk_bprmf_state* st, const hsk_ws& w_all, const hsk_batch_desc& cur, const hsk_batch_desc& n1,
                         const hsk_batch_desc& n2, hipStream_t stream) {
  hsk_aux* a = (hsk_aux*)st->aux;
  int n_part = 1;
  HSK_REQUIRE(hsk_pipe_plan(st, cur.batch, cur.n_neg, &n_part, nullptr, nullptr), HSK_ERR_UNSUPPORTED, "internal: pipeline shape");
  // slots: keep what matches, drop what does not (a wrong guess, a stale run), hand the free ones out
  const hsk_batch_desc* want[3] = {&cur, &n1, &n2};
  int slot_of[3] = {-1, -1, -1};
  bool used[3] = {false, false, false};
  for (int k = 0; k < 3; ++k)
    for (int i = 0; i < 3 && slot_of[k] < 0; ++i)
      if (!used[i] && hsk_pipe_same(a->pipe[i], *want[k])) {
        slot_of[k] = i;
        used[i] = true;
      }
  int rc;
  for (int i = 0; i < 3; ++i)
    if (!used[i] && a->pipe[i].valid && (rc = hsk_pipe_drop(st, w_all, i, stream))) return rc;
  for (int k = 0; k < 3; ++k) {
    if (!want[k]->valid || slot_of[k] >= 0) continue;
    for (int i = 0; i < 3 && slot_of[k] < 0; ++i)
      if (!used[i]) {
        slot_of[k] = i;
        used[i] = true;
        a->pipe[i] = hsk_aux::pipe_slot{true, want[k]->order, want[k]->start, want[k]->batch, want[k]->n_neg, want[k]->step,
                                        HSK_PIPE_NONE};
      }
  }
  // HSK_PIPE_ALONE=1 (experiments): nothing rides -- every phase of the step's batch as a launch of its own in front of it
  // (a timeline then shows the forward, the item/user launch and the five phases each by itself)
  static const int alone = getenv("HSK_PIPE_ALONE") ? atoi(getenv("HSK_PIPE_ALONE")) : 0;
  const int sc = slot_of[0], s1 = alone ? -1 : slot_of[1], s2 = alone ? -1 : slot_of[2];
  if (alone) {
    const hsk_ws w = hsk_select(w_all, sc);
    while (a->pipe[sc].stage < HSK_PIPE_B) {
      if ((rc = hsk_pipe_launch_phase(st, w, a->pipe[sc], a->pipe[sc].stage + 1, stream))) return rc;
      a->pipe[sc].stage += 1;
    }
  }
  // what should have ridden in earlier steps and did not (cold start): on this stream, now
  for (int k = 0; k < 2; ++k) {
    const int sl = k == 0 ? sc : s1;
    if (sl < 0) continue;
    const int need = k == 0 ? HSK_PIPE_C : HSK_PIPE_H;
    const hsk_ws w = hsk_select(w_all, sl);
    while (a->pipe[sl].stage < need) {
      if ((rc = hsk_pipe_launch_phase(st, w, a->pipe[sl], a->pipe[sl].stage + 1, stream))) return rc;
      a->pipe[sl].stage += 1;
    }
  }
  // HSK_PIPE_SOLO=mask (experiments: what a rider costs its host launch): 1 sampler, 2 histogram, 4 row scan, 8 scatter,
  // 16 bucket -- that phase (and the ones in front of it) as a launch of its own in front of the step instead of riding
  static const int solo = getenv("HSK_PIPE_SOLO") ? atoi(getenv("HSK_PIPE_SOLO")) : 0;
  if (solo) {
    const int upto[3] = {(solo & 16) ? HSK_PIPE_B : 0, (solo & 8) ? HSK_PIPE_C : (solo & 4) ? HSK_PIPE_R : 0,
                         (solo & 2) ? HSK_PIPE_H : (solo & 1) ? HSK_PIPE_S : 0};
    const int sl3[3] = {sc, s1, s2};
    for (int k = 0; k < 3; ++k) {
      if (sl3[k] < 0) continue;
      const hsk_ws w = hsk_select(w_all, sl3[k]);
      while (a->pipe[sl3[k]].stage < upto[k]) {
        if ((rc = hsk_pipe_launch_phase(st, w, a->pipe[sl3[k]], a->pipe[sl3[k]].stage + 1, stream))) return rc;
        a->pipe[sl3[k]].stage += 1;
      }
    }
  }
  // what rides in this step's two launches
  hsk_ride_fwd rf = {};
  hsk_ride_item ri = {};
  {
    const hsk_ws wc = hsk_select(w_all, sc);
    if (a->pipe[sc].stage == HSK_PIPE_C) {
      rf.Bk = hsk_pipe_ride_sort(st, wc, a->pipe[sc], HSK_PIPE_B);
      a->pipe[sc].stage = HSK_PIPE_B;
    }
  }
  if (s1 >= 0) {
    const hsk_ws w1 = hsk_select(w_all, s1);
    if (a->pipe[s1].stage == HSK_PIPE_H) {
      rf.R = hsk_pipe_ride_sort(st, w1, a->pipe[s1], HSK_PIPE_R);
      a->pipe[s1].stage = HSK_PIPE_R;
    }
    if (a->pipe[s1].stage == HSK_PIPE_R) {
      ri.C = hsk_pipe_ride_sort(st, w1, a->pipe[s1], HSK_PIPE_C);
      a->pipe[s1].stage = HSK_PIPE_C;
    }
  }
  if (s2 >= 0) {
    const hsk_ws w2 = hsk_select(w_all, s2);
    const hsk_aux::pipe_slot& b = a->pipe[s2];
    if (b.stage == HSK_PIPE_NONE) {
      static const int s_per_wave = getenv("HSK_PIPE_S_PER_WAVE") ? std::min(64, std::max(1, atoi(getenv("HSK_PIPE_S_PER_WAVE")))) : 8;
      rf.S = hsk_ride_sample{(int)hsk_ceil_div(b.batch, 4 * s_per_wave), st->coo_user, st->coo_item, b.order, (long long)b.start,
                             (int)b.batch, (int)b.n_neg, st->csr_indptr, st->csr_indices, (int)st->n_items, st->seed,
                             (uint64_t)b.step, w2.u32, w2.it32, w2.owner, w2.cnt, st->status,
                             hsk_alias{st->alias_prob, st->alias_idx}, w2.stamp, n_part, s_per_wave};
      a->pipe[s2].stage = HSK_PIPE_S;
    }
    if (a->pipe[s2].stage == HSK_PIPE_S) {
      ri.H = hsk_pipe_ride_sort(st, w2, a->pipe[s2], HSK_PIPE_H);
      a->pipe[s2].stage = HSK_PIPE_H;
    }
  }
  rf.n_total = (int)hsk_align_up(rf.Bk.n_blocks + rf.S.n_blocks + rf.R.n_blocks, 8);
  ri.n_total = (int)hsk_align_up(ri.C.n_blocks + ri.H.n_blocks, 8);
  // the ahead-of-time replay of the next batch's lazily updated user rows takes the next batch from the hint
  a->pf_valid = false;
  a->hint_valid = n1.valid;
  a->hint_order = n1.order;
  a->hint_start = n1.start;
  a->hint_batch = n1.batch;
  a->hint_nneg = n1.n_neg;
  a->pipe_active = true;
  a->ride_fwd = &rf;
  a->ride_item = &ri;
  rc = hsk_run_step(st, w_all, sc, true, cur.batch, cur.n_neg + 1, stream, n_part);
  a->pipe_active = false;
  a->ride_fwd = nullptr;
  a->ride_item = nullptr;
  a->hint_valid = false;
  a->pipe[sc] = hsk_aux::pipe_slot{};   // trained on: its owner map was given back by the user update
  a->pipe_steps += 1;
  return rc;
}

extern "C" int hsk_bprmf_train_step(hsk_bprmf_state* st, const int64_t* u_idx, const int64_t* i_idx, int64_t batch,
                                    int64_t n_cols, hsk_stream_t stream_) {
  int rc = hsk_check_state(st);
  if (rc) return rc;
  HSK_REQUIRE(u_idx && i_idx, HSK_ERR_INVALID, "u_idx / i_idx must not be NULL");
  if ((rc = hsk_check_batch(st, batch, n_cols))) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  hsk_ws w = hsk_carve_st(st);
  const int64_t total = batch * n_cols;
  st->timing_now = st->timing && (st->timing_every <= 1 || ((st->step + 1) % st->timing_every) == 0);
  if ((rc = hsk_pipe_reset(st, w, stream))) return rc;
  if ((rc = hsk_discard_prefetch(st, w, stream))) return rc;
  const int set = st->aux ? (((hsk_aux*)st->aux)->cur_set ^ 1) : 0;
  const hsk_ws ws = hsk_select(w, set);
  const int n_part = hsk_part_rule_st(st, batch, n_cols - 1);
  HSK_STAGE(HSK_STAGE_PREP, k_prep_external<<<(unsigned)hsk_ceil_div(total, 256), 256, 0, stream>>>(
                                u_idx, i_idx, (int)batch, (int)n_cols, (int)st->n_users, (int)st->n_items, ws.u32,
                                ws.it32, ws.owner, ws.cnt, st->status, ws.stamp, (int)st->step + 1, n_part));
  HSK_LAUNCH_CHECK();
  return hsk_run_step(st, w, set, false, batch, n_cols, stream, n_part);
}

extern "C" int hsk_bprmf_train_step_sampled(hsk_bprmf_state* st, const int64_t* order, int64_t start, int64_t batch,
                                            int64_t n_neg, hsk_stream_t stream_) {
  int rc = hsk_check_state(st);
  if (rc) return rc;
  HSK_REQUIRE(st->csr_indptr && st->csr_indices && st->coo_user && st->coo_item, HSK_ERR_INVALID,
              "CSR/COO of the training interactions missing from the state");
  if ((rc = hsk_check_batch(st, batch, n_neg + 1))) return rc;
  HSK_REQUIRE(start >= 0 && start + batch <= st->nnz, HSK_ERR_INVALID, "interaction range [%lld, %lld) outside nnz %lld",
              (long long)start, (long long)(start + batch), (long long)st->nnz);
  hipStream_t stream = (hipStream_t)stream_;
  hsk_ws w = hsk_carve_st(st);
  // the RNG stream id is the index of the step about to be taken: every step draws fresh negatives
  st->timing_now = st->timing && (st->timing_every <= 1 || ((st->step + 1) % st->timing_every) == 0);
  hsk_aux* aux = (hsk_aux*)st->aux;
  const int n_part = hsk_part_rule_st(st, batch, n_neg);
  if ((rc = hsk_pipe_reset(st, w, stream))) return rc;
  if (aux && aux->pf_valid && aux->pf_order == order && aux->pf_start == start && aux->pf_batch == batch &&
      aux->pf_nneg == n_neg && aux->pf_step == st->step) {
    // this batch was sampled and sorted on the side stream during the previous step
    aux->pf_valid = false;
    HSK_HIP(hipStreamWaitEvent(stream, aux->ev_ready, 0));
    return hsk_run_step(st, w, aux->cur_set ^ 1, true, batch, n_neg + 1, stream, n_part);
  }
  if ((rc = hsk_discard_prefetch(st, w, stream))) return rc;
  const int set = aux ? (aux->cur_set ^ 1) : 0;
  if ((rc = hsk_launch_prep_sample(st, hsk_select(w, set), order, start, batch, n_neg, (uint64_t)st->step, stream)))
    return rc;
  return hsk_run_step(st, w, set, false, batch, n_neg + 1, stream, n_part);
}

// =============================================================================================
// replayed HIP graphs: the steady-state inner loop without the host
// =============================================================================================
// At the reference's usual batch sizes (128..512) a step is a handful of 3-20 us kernels and the host's ~10 HIP calls
// per step (3-5 us each) are what bounds it.  A run of `chunk` consecutive steps is therefore CAPTURED once -- the same
// launch sequence as the eager path, two streams, fork / join by events, the next batch prepared under the current
// one -- and replayed with one hipGraphLaunch per chunk.  Whatever differs between two replays (batch offset, step
// index -> RNG stream id and Adam bias corrections, permutation pointer) is read by the kernels from a device
// descriptor (hsk_step_desc) that a one-thread kernel rewrites in front of each replay.  Same kernels, same order,
// same arithmetic: results are bit-identical to the eager sequence.
__global__ void k_set_desc(hsk_step_desc* d, long long start0, const int64_t* order, int step0) {
  d->start0 = start0;
  d->order = order;
  d->step0 = step0;
  d->pad = 0;
}

#define HSK_GRAPH_DEFAULT_CHUNK 64
#define HSK_GRAPH_MAX_CACHED 8

// What a captured graph has frozen into its kernel arguments: every pointer of the state (tables, moments, biases,
// CSR / COO, alias table, workspace, loss_out, status), the shapes, the hyper-parameters (lr, wd, betas, eps -> the
// optimiser scalars and the Adam table), seed, loss / optimiser kind, the lazy flags.  Only the step counter and the
// timing fields may differ between capture and replay (the former is read from the device descriptor).  A caller that
// changes anything else between two hsk_bprmf_train_steps calls -- an LR schedule through st->lr, parameters rebound
// after .to(), a new dataset -- gets a fresh capture, exactly as the eager path would pick the new values up.
static_assert(sizeof(hsk_bprmf_state) == 424, "hsk_bprmf_state has no padding holes: its byte image is a valid key");
static hsk_bprmf_state hsk_graph_key(const hsk_bprmf_state* st) {
  hsk_bprmf_state k;
  memcpy(&k, st, sizeof(k));
  k.step = 0;
  k.timing = nullptr;
  k.timing_mask = k.timing_every = k.timing_now = 0;
  return k;
}

static int64_t hsk_graph_chunk(const hsk_bprmf_state* st) {
  static const int env = getenv("HSK_GRAPH") ? atoi(getenv("HSK_GRAPH")) : 1;
  if (!env || st->graph_chunk < 0) return 0;
  const hsk_aux* a = (const hsk_aux*)st->aux;
  if (!a || a->graph_broken) return 0;
  if (st->timing && st->timing_mask) return 0;            // stage timing needs events between the launches
  if (st->dim % 2 != 0) return 0;                         // the merged item + user launch carries the descriptor
  if ((st->lazy_users || st->lazy_items) && !hsk_adam_tab_saturates(st)) return 0;
  if (st->step + 2 * HSK_GRAPH_DEFAULT_CHUNK >= HSK_ADAM_TAB_LEN && !hsk_adam_tab_saturates(st)) return 0;
  static const int env_chunk = getenv("HSK_GRAPH_CHUNK") ? atoi(getenv("HSK_GRAPH_CHUNK")) : 0;   // experiments
  return st->graph_chunk > 0 ? st->graph_chunk : env_chunk > 1 ? env_chunk : HSK_GRAPH_DEFAULT_CHUNK;
}

// capture `n` steps of `batch` positives starting (relative to the descriptor) at step 0, first buffer set `set0`
static int hsk_capture_steps(hsk_bprmf_state* st, const hsk_ws& w, int64_t n, int64_t batch, int64_t n_neg, int set0,
                             hipGraphExec_t* out) {
  hsk_aux* aux = (hsk_aux*)st->aux;
  hipStream_t stream = aux->cap;
  const int64_t K = n_neg + 1, total = batch * K;
  if (hsk_sort_lds_fits(st->n_items, total)) {
    if (hsk_sort_lds_bytes(st->n_items) > 65536)
      HSK_HIP(hipFuncSetAttribute((const void*)k_sort_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)hsk_sort_lds_bytes(st->n_items)));
  } else if (total > 1024 * 8) {   // dynamic-LDS opt-in of the bucket sort: not a stream operation, do it before the capture
    hsk_sort_plan plan;
    HSK_REQUIRE(hsk_make_sort_plan((int)st->n_items, total, &plan) == 0, HSK_ERR_UNSUPPORTED, "item sort: n_items too large");
    const size_t bucket_lds = ((size_t)(st->lazy_items ? 6 : 5) * plan.ipb + 2) * sizeof(int);
    if (bucket_lds > 65536)
      HSK_HIP(hipFuncSetAttribute((const void*)k_sort_bucket, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bucket_lds));
  }
  const int64_t step_saved = st->step;
  const int set_saved = aux->cur_set, slot_saved = aux->cur_slot;
  const int timing_saved = st->timing_now;
  st->timing_now = 0;
  aux->g_desc = w.desc;
  int rc = HSK_OK;
  hipGraph_t graph = nullptr;
  if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    aux->g_desc = nullptr;
    hsk_set_error("hipStreamBeginCapture failed");
    return HSK_ERR_HIP;
  }
  int set = set0;
  // Grouped preparation: the batches of G consecutive steps are sampled and sorted by one launch each -- group 0 on the
  // main stream at the head of the run, group i+1 on the side stream from the first step of group i on; the steps
  // themselves launch no prefetch.  (k_sort_lds shapes only: one workgroup per batch.)
  const int G = (!st->lazy_items && hsk_sort_lds_fits(st->n_items, total)) ? std::min<int64_t>(w.group, n) : 1;
  auto prepare_group = [&](int gset, int64_t first, hipStream_t q) -> int {
    const hsk_ws wg = hsk_select(w, gset);
    const int cnt_g = (int)std::min<int64_t>(G, n - first);
    k_prep_sample_group<<<(unsigned)hsk_ceil_div((int64_t)cnt_g * batch, 4), 256, 0, q>>>(
        st->coo_user, st->coo_item, (int)batch, (int)n_neg, st->csr_indptr, st->csr_indices, (int)st->n_items, st->seed,
        wg.u32, wg.it32, wg.owner, wg.cnt, st->status, hsk_alias{st->alias_prob, st->alias_idx}, w.desc, (int)first,
        wg.stamp, cnt_g, (long long)w.gs_batch, (long long)w.gs_ent, (long long)w.gs_users);
    k_sort_lds<<<(unsigned)cnt_g, 1024, hsk_sort_lds_bytes(st->n_items), q>>>(
        wg.it32, (int)total, (int)st->n_items, wg.perm, wg.offsets, nullptr, nullptr, nullptr, (long long)w.gs_ent,
        (long long)w.gs_items);
    return hipGetLastError() == hipSuccess ? HSK_OK : HSK_ERR_HIP;
  };
  if (G > 1) {
    aux->grouped = true;
    for (int64_t s = 0; s < n && rc == HSK_OK; ++s) {
      const int64_t gi = s / G;
      const int slot = (int)(s - gi * G);
      const int gset = (set0 + (int)gi) & 1;
      aux->g_rel = (int)s;
      if (s == 0) rc = prepare_group(gset, 0, stream);
      if (rc == HSK_OK && slot == 0) {
        if (s > 0 && hipStreamWaitEvent(stream, aux->ev_ready, 0) != hipSuccess) rc = HSK_ERR_HIP;   // this group is ready
        if (rc == HSK_OK && (gi + 1) * G < n) {   // the next group: on the side stream from here on
          if (hipEventRecord(aux->ev_fork, stream) != hipSuccess || hipStreamWaitEvent(aux->side, aux->ev_fork, 0) != hipSuccess)
            rc = HSK_ERR_HIP;
          if (rc == HSK_OK) rc = prepare_group(gset ^ 1, (gi + 1) * G, aux->side);
          if (rc == HSK_OK && hipEventRecord(aux->ev_ready, aux->side) != hipSuccess) rc = HSK_ERR_HIP;
        }
      }
      if (rc == HSK_OK && s + 1 < n) {   // names the next batch for the ahead-of-time user catch-up (no prefetch: grouped)
        aux->hint_valid = true;
        aux->hint_order = nullptr;
        aux->hint_start = 0;
        aux->hint_batch = batch;
        aux->hint_nneg = n_neg;
      } else {
        aux->hint_valid = false;
      }
      if (rc == HSK_OK) rc = hsk_run_step(st, w, gset, true, batch, K, stream, 1, slot);
    }
    aux->grouped = false;
  }
  for (int64_t s = 0; s < n && rc == HSK_OK && G <= 1; ++s) {
    aux->g_rel = (int)s;
    bool sorted = true;
    if (s == 0) {   // the run's first batch is prepared inside the graph, on the main stream
      rc = hsk_launch_prep_sample(st, hsk_select(w, set), nullptr, 0, batch, n_neg, 0, stream);
      sorted = false;
    } else {
      if (hipStreamWaitEvent(stream, aux->ev_ready, 0) != hipSuccess) rc = HSK_ERR_HIP;
      aux->pf_valid = false;
    }
    if (rc == HSK_OK && s + 1 < n) {   // name the next batch: prepared on the side stream under this step
      aux->hint_valid = true;
      aux->hint_order = nullptr;
      aux->hint_start = 0;
      aux->hint_batch = batch;
      aux->hint_nneg = n_neg;
    }
    if (rc == HSK_OK)
      rc = hsk_run_step(st, w, set, sorted, batch, K, stream,
                        hsk_part_rule_st(st, batch, n_neg));
    set ^= 1;
  }
  const hipError_t e = hipStreamEndCapture(stream, &graph);
  aux->g_desc = nullptr;
  aux->g_rel = 0;
  aux->hint_valid = false;
  aux->pf_valid = false;
  aux->cur_set = set_saved;
  aux->cur_slot = slot_saved;
  st->step = step_saved;
  st->timing_now = timing_saved;
  if (rc == HSK_OK && e != hipSuccess) {
    hsk_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
    rc = HSK_ERR_HIP;
  }
  if (rc == HSK_OK && hipGraphInstantiate(out, graph, nullptr, nullptr, 0) != hipSuccess) {
    hsk_set_error("hipGraphInstantiate failed");
    rc = HSK_ERR_HIP;
  }
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipGetLastError();
  return rc;
}

extern "C" int hsk_bprmf_train_steps(hsk_bprmf_state* st, const int64_t* order, int64_t start, int64_t n_steps,
                                     int64_t batch, int64_t n_neg, hsk_stream_t stream_) {
  HSK_REQUIRE(st != nullptr, HSK_ERR_INVALID, "state is NULL");
  HSK_REQUIRE(n_steps >= 0 && batch > 0 && start >= 0 && start + n_steps * batch <= st->nnz, HSK_ERR_INVALID,
              "steps [%lld, +%lld x %lld) outside nnz %lld", (long long)start, (long long)n_steps, (long long)batch,
              (long long)st->nnz);
  int64_t s = 0;
  if (st->aux && n_steps > 0 && hsk_pipe_eligible(st, batch, n_neg)) {
    // large batches on the item-partitioned forward: the preparation of the next two batches rides in the steps' own
    // launches (hsk_pipe_step); the batches behind the run come from hsk_bprmf_hint_after_run[_n]
    int rc = hsk_check_state(st);
    if (rc) return rc;
    HSK_REQUIRE(st->csr_indptr && st->csr_indices && st->coo_user && st->coo_item, HSK_ERR_INVALID,
                "CSR/COO of the training interactions missing from the state");
    if ((rc = hsk_check_batch(st, batch, n_neg + 1))) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    hsk_aux* aux = (hsk_aux*)st->aux;
    const hsk_ws w = hsk_carve_st(st);
    if ((rc = hsk_discard_prefetch(st, w, stream))) return rc;
    aux->hint_valid = false;
    const bool tail_ok = aux->tail_valid && aux->tail_batch == batch && aux->tail_nneg == n_neg;
    auto desc = [&](int64_t k) -> hsk_batch_desc {   // the batch trained on k steps from the start of this run
      if (k < n_steps) return hsk_batch_desc{true, order, start + k * batch, batch, n_neg, 0};
      const int64_t j = k - n_steps;
      if (tail_ok && j < aux->tail_count) return hsk_batch_desc{true, aux->tail_order, aux->tail_start + j * batch, batch, n_neg, 0};
      return hsk_batch_desc{false, nullptr, 0, 0, 0, 0};
    };
    for (; s < n_steps; ++s) {
      hsk_batch_desc cur = desc(s), n1 = desc(s + 1), n2 = desc(s + 2);
      cur.step = st->step;
      n1.step = st->step + 1;
      n2.step = st->step + 2;
      st->timing_now = st->timing && (st->timing_every <= 1 || ((st->step + 1) % st->timing_every) == 0);
      if ((rc = hsk_pipe_step(st, w, cur, n1, n2, stream))) return rc;
    }
    aux->tail_valid = false;
    aux->tail_count = 0;
    return HSK_OK;
  }
  const int64_t chunk_max = hsk_graph_chunk(st);
  if (chunk_max >= 2 && n_steps >= chunk_max) {
    int rc = hsk_check_state(st);
    if (rc) return rc;
    HSK_REQUIRE(st->csr_indptr && st->csr_indices && st->coo_user && st->coo_item, HSK_ERR_INVALID,
                "CSR/COO of the training interactions missing from the state");
    if ((rc = hsk_check_batch(st, batch, n_neg + 1))) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    hsk_aux* aux = (hsk_aux*)st->aux;
    const hsk_ws w = hsk_carve_st(st);
    const int flags = (st->lazy_users ? 1 : 0) | (st->loss_kind << 1) | (st->opt_kind << 4) | (st->lazy_items ? 64 : 0);
    while (n_steps - s >= chunk_max) {
      const int64_t n = chunk_max;
      if ((rc = hsk_discard_prefetch(st, w, stream))) return rc;
      aux->hint_valid = false;
      const int set0 = 0;   // the buffer sets are scratch: with no prepared batch pending a run may start with either
      hipGraphExec_t exec = nullptr;
      const hsk_bprmf_state key = hsk_graph_key(st);
      for (size_t gi = 0; gi < aux->graphs.size();) {
        auto& g = aux->graphs[gi];
        if (g.n_steps == n && g.batch == batch && g.n_neg == n_neg && g.set0 == set0 && g.flags == flags) {
          if (memcmp(&g.key, &key, sizeof(key)) == 0) {
            exec = g.exec;
          } else {   // same shape, another state: the captured arguments are stale
            (void)hipGraphExecDestroy(g.exec);
            aux->graphs.erase(aux->graphs.begin() + gi);
            continue;
          }
        }
        ++gi;
      }
      if (!exec) {
        static const int dbg = getenv("HSK_DEBUG_GRAPH") ? atoi(getenv("HSK_DEBUG_GRAPH")) : 0;
        if (dbg) fprintf(stderr, "hsk: capturing %lld steps (batch %lld, %zu cached)\n", (long long)n, (long long)batch, aux->graphs.size());
        if ((rc = hsk_capture_steps(st, w, n, batch, n_neg, set0, &exec))) {
          aux->graph_broken = true;   // eager launches from here on (the error text stays available)
          break;
        }
        if (aux->graphs.size() >= HSK_GRAPH_MAX_CACHED) {
          (void)hipGraphExecDestroy(aux->graphs.front().exec);
          aux->graphs.erase(aux->graphs.begin());
        }
        aux->graphs.push_back({exec, n, batch, n_neg, set0, flags, key});
      }
      k_set_desc<<<1, 1, 0, stream>>>(w.desc, (long long)(start + s * batch), order, (int)st->step);
      HSK_LAUNCH_CHECK();
      HSK_HIP(hipGraphLaunch(exec, stream));
      aux->graph_launches += 1;
      st->step += n;
      {
        const int Gr = (!st->lazy_items && hsk_sort_lds_fits(st->n_items, batch * (n_neg + 1))) ? (int)std::min<int64_t>(w.group, n) : 1;
        if (Gr > 1) {   // grouped preparation: the last batch sits in slot (n-1) % G of set (set0 + (n-1)/G) & 1
          aux->cur_set = (set0 + (int)((n - 1) / Gr)) & 1;
          aux->cur_slot = (int)((n - 1) % Gr);
        } else {
          aux->cur_set = (n & 1) ? set0 : (set0 ^ 1);
          aux->cur_slot = 0;
        }
        aux->last_set = aux->cur_set;
        aux->last_slot = aux->cur_slot;
      }
      s += n;
      // lazily updated rows: the periodic sweeps that came due during the replayed run follow it (the cadence is a speed
      // matter, any replay length is exact)
      if ((st->lazy_users || st->lazy_items) &&
          (rc = hsk_periodic_flush(st, w, stream, st->step - n, (double)batch, (double)(batch * (n_neg + 1)))))
        return rc;
    }
  }
  hsk_aux* tail_aux = (hsk_aux*)st->aux;
  for (; s < n_steps; ++s) {
    if (st->aux && s + 1 < n_steps) {
      int hrc = hsk_bprmf_hint_next(st, order, start + (s + 1) * batch, batch, n_neg);
      if (hrc) return hrc;
    } else if (tail_aux && tail_aux->tail_valid) {   // the run's last step prepares the batch the caller named
      int hrc = hsk_bprmf_hint_next(st, tail_aux->tail_order, tail_aux->tail_start, tail_aux->tail_batch, tail_aux->tail_nneg);
      if (hrc) return hrc;
    }
    int rc = hsk_bprmf_train_step_sampled(st, order, start + s * batch, batch, n_neg, stream_);
    if (rc) return rc;
  }
  if (tail_aux) tail_aux->tail_valid = false;
  return HSK_OK;
}

extern "C" int64_t hsk_bprmf_pipelined_steps(const hsk_bprmf_state* st) {
  const hsk_aux* a = st ? (const hsk_aux*)st->aux : nullptr;
  return a ? a->pipe_steps : 0;
}

extern "C" int64_t hsk_bprmf_graph_replays(const hsk_bprmf_state* st) {
  const hsk_aux* a = st ? (const hsk_aux*)st->aux : nullptr;
  return a ? a->graph_launches : 0;
}

extern "C" int hsk_bprmf_flush(hsk_bprmf_state* st, hsk_stream_t stream_) {
  int rc = hsk_check_state(st);
  if (rc) return rc;
  hsk_ws w = hsk_carve_st(st);
  // a flush ends a run of steps (epoch end, evaluation, checkpoint): a hinted or prefetched batch that was never
  // trained on must not survive it -- the next epoch's permutation may well be allocated at the same address
  if (st->aux) {
    ((hsk_aux*)st->aux)->hint_valid = false;
    if ((rc = hsk_discard_prefetch(st, w, (hipStream_t)stream_))) return rc;
    if ((rc = hsk_pipe_reset(st, w, (hipStream_t)stream_))) return rc;
  }
  if (!st->lazy_users && !st->lazy_items) return HSK_OK;  // dense updates: nothing is pending
  return hsk_launch_flush(st, w, (hipStream_t)stream_);
}

extern "C" int hsk_bprmf_last_batch(const hsk_bprmf_state* st, int64_t batch, int64_t n_cols, int64_t* u_out,
                                    int64_t* i_out, hsk_stream_t stream_) {
  int rc = hsk_check_state(st);
  if (rc) return rc;
  HSK_REQUIRE(u_out && i_out, HSK_ERR_INVALID, "output pointers must not be NULL");
  if ((rc = hsk_check_batch(st, batch, n_cols))) return rc;
  hsk_ws w = hsk_carve_st(st);
  const int64_t total = batch * n_cols;
  if (st->aux) w = hsk_select(w, ((hsk_aux*)st->aux)->last_set, ((hsk_aux*)st->aux)->last_slot);
  // partitioned row layout: the positive sits in n_part columns, the caller sees it once
  const int n_part = hsk_part_rule_st(st, batch, n_cols - 1);
  k_widen_batch<<<(unsigned)hsk_ceil_div(total, 256), 256, 0, (hipStream_t)stream_>>>(w.u32, w.it32, batch, total,
                                                                                      u_out, i_out, (int)n_cols, n_part);
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

extern "C" int64_t hsk_bprmf_batch_columns(const hsk_bprmf_state* st, int64_t batch, int64_t n_cols) {
  if (!st || batch <= 0 || n_cols < 2) return -1;
  return hsk_part_cols(n_cols, hsk_part_rule_st(st, batch, n_cols - 1));
}

extern "C" int hsk_bprmf_last_sort(const hsk_bprmf_state* st, int64_t n_entries, int32_t* perm_out, int32_t* offsets_out,
                                   hsk_stream_t stream_) {
  int rc = hsk_check_state(st);
  if (rc) return rc;
  HSK_REQUIRE(perm_out && offsets_out && n_entries > 0 && n_entries <= st->max_batch * (st->max_cols + HSK_PART_MAX - 1),
              HSK_ERR_INVALID, "bad argument");
  hsk_ws w = hsk_carve_st(st);
  if (st->aux) w = hsk_select(w, ((hsk_aux*)st->aux)->last_set, ((hsk_aux*)st->aux)->last_slot);
  HSK_HIP(hipMemcpyAsync(perm_out, w.perm, n_entries * sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream_));
  HSK_HIP(hipMemcpyAsync(offsets_out, w.offsets, (st->n_items + 1) * sizeof(int), hipMemcpyDeviceToDevice,
                         (hipStream_t)stream_));
  return HSK_OK;
}

extern "C" int hsk_sample_negatives_alias(const int64_t* csr_indptr, const int32_t* csr_indices, int64_t n_users,
                                          int64_t n_items, const float* alias_prob, const int32_t* alias_idx,
                                          const int64_t* u_idx, int64_t batch, int64_t n_neg, uint64_t seed,
                                          uint64_t stream_id, int64_t* neg_out, int32_t* status,
                                          hsk_stream_t stream_) {
  HSK_REQUIRE(csr_indptr && csr_indices && u_idx && neg_out && alias_prob && alias_idx, HSK_ERR_INVALID,
              "NULL pointer argument");
  HSK_REQUIRE(n_users > 0 && n_items > 0 && n_items < 0x7fffffff && n_users < 0x7fffffff, HSK_ERR_INVALID,
              "bad n_users / n_items");
  HSK_REQUIRE(batch >= 0 && n_neg >= 1, HSK_ERR_INVALID, "bad batch / n_neg");
  if (batch == 0) return HSK_OK;
  k_sample_negatives<<<(unsigned)hsk_ceil_div(batch, 4), 256, 0, (hipStream_t)stream_>>>(
      csr_indptr, csr_indices, (int)n_users, (int)n_items, u_idx, (int)batch, (int)n_neg, seed, stream_id, neg_out,
      status, hsk_alias{alias_prob, alias_idx});
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

// multi-GPU phases (item table range-sharded, user table row-sharded)
#include <algorithm>
#include "hsk_shard.inc"
#include "hsk_rccl.inc"

// =============================================================================================
// Generic embedding gather + its dense backward (the gather primitive the reference's other SGD models share:
// nn.Embedding in ACF / UProtoMF / IProtoMF / UIProtoMF, algorithms/sgd_alg.py:187-570).  Forward: out[j] =
// table[idx[j]].  Backward: grad_table[r] = sum of grad_out[j] over the positions j with idx[j] == r, added in
// ascending j (the deterministic item sort of this file builds the row-major index), zero for rows nobody named --
// what autograd's embedding_dense_backward returns, without atomics.
// =============================================================================================
template <int V, int NCH, bool FULL>
__global__ __launch_bounds__(256) void k_rows_gather(const float* __restrict__ table, int n_rows, int D,
                                                     const int64_t* __restrict__ idx, int n, float* __restrict__ out,
                                                     int32_t* status) {
  const int lane = hsk_lane();
  const int j = blockIdx.x * 4 + hsk_uniform_i(threadIdx.x >> 6);
  if (j >= n) return;
  const int r = hsk_uniform_i(hsk_clamp_index(idx[j], n_rows, status));
  hsk_row<V, NCH> row;
  hsk_row_load<V, NCH, FULL>(row, table + (long long)r * D, lane, D);
  hsk_row_store<V, NCH, FULL>(row, out + (long long)j * D, lane, D);
}

__global__ __launch_bounds__(256) void k_idx_to_i32(const int64_t* __restrict__ idx, int n, int n_rows,
                                                    int* __restrict__ it32, int32_t* status) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) it32[j] = hsk_clamp_index(idx[j], n_rows, status);
}

template <int V, int NCH, bool FULL>
__global__ __launch_bounds__(256) void k_rows_segment_sum(const float* __restrict__ grad_out, const int* __restrict__ perm,
                                                          const int* __restrict__ offsets, int n_rows, int D,
                                                          float* __restrict__ grad_table) {
  const int lane = hsk_lane();
  const int r = blockIdx.x * 4 + hsk_uniform_i(threadIdx.x >> 6);
  if (r >= n_rows) return;
  using Row = hsk_row<V, NCH>;
  const int beg = hsk_uniform_i(offsets[r]), end = hsk_uniform_i(offsets[r + 1]);
  Row acc;
  hsk_row_zero(acc);
  for (int c0 = beg; c0 < end; c0 += 64) {
    const int nr = min(64, end - c0);
    const int mye = (lane < nr) ? perm[c0 + lane] : 0;
    for (int j = 0; j < nr; j += 4) {
      Row buf[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (j + q < nr)
          hsk_row_load<V, NCH, FULL>(buf[q], grad_out + (long long)hsk_readlane_i(mye, min(j + q, 63)) * D, lane, D);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (j + q < nr) hsk_row_add(acc, buf[q]);
    }
  }
  hsk_row_store<V, NCH, FULL>(acc, grad_table + (long long)r * D, lane, D);
}

extern "C" int hsk_embedding_gather(const float* table, int64_t n_rows, int64_t dim, const int64_t* idx, int64_t n,
                                    float* out, int32_t* status, hsk_stream_t stream_) {
  HSK_REQUIRE(table && idx && out, HSK_ERR_INVALID, "NULL pointer argument");
  HSK_REQUIRE(n_rows > 0 && n_rows < 0x7fffffff && dim > 0 && n >= 0 && n < 0x7fffffff, HSK_ERR_INVALID, "bad sizes");
  if (n == 0) return HSK_OK;
  hipStream_t stream = (hipStream_t)stream_;
  int rc = hsk_dispatch_dim(dim, [&](auto v_, auto n_, auto f_) {
    constexpr int V = decltype(v_)::value;
    constexpr int NCH = decltype(n_)::value;
    constexpr bool FULL = decltype(f_)::value;
    k_rows_gather<V, NCH, FULL><<<(unsigned)hsk_ceil_div(n, 4), 256, 0, stream>>>(table, (int)n_rows, (int)dim, idx,
                                                                                  (int)n, out, status);
    return HSK_OK;
  });
  if (rc) return rc;
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

struct hsk_embw {   // scratch of hsk_embedding_backward
  int* it32;
  int2* perm1;
  int *perm, *hist, *btot, *bstart, *offsets;
  int64_t total;
};

static hsk_embw hsk_embw_carve(void* base, int64_t n_rows, int64_t n) {
  hsk_embw w;
  char* p = (char*)base;
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    char* r = p ? p + off : nullptr;
    off += hsk_align_up(bytes, 256);
    return r;
  };
  const int64_t hist_elems = hsk_sort_hist_elems(n_rows, std::max<int64_t>(n, 1));
  w.it32 = (int*)take(n * 4);
  w.perm1 = (int2*)take(n * 8);
  w.perm = (int*)take(n * 4);
  w.hist = (int*)take((hist_elems > 0 ? hist_elems : 4) * 4);
  w.btot = (int*)take(HSK_SORT_MAX_BUCKETS * 4);
  w.bstart = (int*)take((HSK_SORT_MAX_BUCKETS + 1) * 4);
  w.offsets = (int*)take((n_rows + 1) * 4);
  w.total = off;
  return w;
}

extern "C" int64_t hsk_embedding_backward_ws_bytes(int64_t n_rows, int64_t n) {
  if (n_rows <= 0 || n <= 0 || n >= 0x7fffffff || hsk_sort_hist_elems(n_rows, n) < 0) return -1;
  return hsk_embw_carve(nullptr, n_rows, n).total;
}

extern "C" int hsk_embedding_backward(const float* grad_out, const int64_t* idx, int64_t n, int64_t n_rows, int64_t dim,
                                      float* grad_table, void* ws, int64_t ws_bytes, int32_t* status,
                                      hsk_stream_t stream_) {
  HSK_REQUIRE(grad_out && idx && grad_table && ws, HSK_ERR_INVALID, "NULL pointer argument");
  HSK_REQUIRE(n_rows > 0 && dim > 0 && n > 0, HSK_ERR_INVALID, "bad sizes");
  const int64_t need = hsk_embedding_backward_ws_bytes(n_rows, n);
  HSK_REQUIRE(need > 0, HSK_ERR_UNSUPPORTED, "table of %lld rows too large for the index sort", (long long)n_rows);
  HSK_REQUIRE(ws_bytes >= need && ((uintptr_t)ws & 255) == 0, HSK_ERR_INVALID, "workspace: %lld bytes needed, %lld given",
              (long long)need, (long long)ws_bytes);
  hipStream_t stream = (hipStream_t)stream_;
  const hsk_embw e = hsk_embw_carve(ws, n_rows, n);
  k_idx_to_i32<<<(unsigned)hsk_ceil_div(n, 256), 256, 0, stream>>>(idx, (int)n, (int)n_rows, e.it32, status);
  HSK_LAUNCH_CHECK();
  // the item sort of the fused step, on a throw-away state that only says "n_rows keys, nothing lazy, no timing"
  hsk_bprmf_state fake;
  memset(&fake, 0, sizeof(fake));
  fake.n_items = n_rows;
  hsk_ws w;
  memset(&w, 0, sizeof(w));
  w.it32 = e.it32;
  w.perm1 = e.perm1;
  w.perm = e.perm;
  w.hist = e.hist;
  w.btot = e.btot;
  w.bstart = e.bstart;
  w.offsets = e.offsets;
  hsk_bprmf_state* st = &fake;   // HSK_STAGE reads st->timing (NULL here)
  (void)st;
  int rc = hsk_launch_sort(&fake, w, n, stream);
  if (rc) return rc;
  rc = hsk_dispatch_dim(dim, [&](auto v_, auto n_, auto f_) {
    constexpr int V = decltype(v_)::value;
    constexpr int NCH = decltype(n_)::value;
    constexpr bool FULL = decltype(f_)::value;
    k_rows_segment_sum<V, NCH, FULL><<<(unsigned)hsk_ceil_div(n_rows, 4), 256, 0, stream>>>(
        grad_out, e.perm, e.offsets, (int)n_rows, (int)dim, grad_table);
    return HSK_OK;
  });
  if (rc) return rc;
  HSK_LAUNCH_CHECK();
  return HSK_OK;
}

#ifdef HSK_DEBUG_XCC
// debugging builds only (make EXTRA=-DHSK_DEBUG_XCC): [workgroup label][XCC id] counts of k_item_user's item workgroups
extern "C" int hsk_debug_xcc(unsigned* out64, int reset) {
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(hsk_dbg_xcc), 64 * sizeof(unsigned)) != hipSuccess) return -1;
  if (reset) {
    unsigned z[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(hsk_dbg_xcc), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
