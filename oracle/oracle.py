"""TEST INFRASTRUCTURE ONLY -- Python face of the CPU oracle (oracle/bprmf_oracle.c) plus the numpy
restatement of the reference's host-side sampler.  Import from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg only; never from hassaku_amd/.

Parity status: PINNED by golden vectors generated from the imported reference
(oracle/gen_golden.py -> tests/golden/*.npz, checked in tests/test_oracle_golden.py).
"""
import ctypes
import os
import subprocess
from ctypes import c_double, c_int32, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, 'liboracle.so')
# sanitizer build of the same source (-fsanitize host build of the CPU restatement): built by
# build_sanitized(), loaded instead of liboracle.so when HSK_ORACLE_SO names it (tests/test_oracle_sanitized.py runs
# the golden-vector tests against it in a child process with libasan preloaded)
SAN_SO_PATH = os.path.join(_HERE, 'liboracle_san.so')
_lib = None


def build(force: bool = False):
    src = os.path.join(_HERE, 'bprmf_oracle.c')
    if force or not os.path.isfile(SO_PATH) or os.path.getmtime(SO_PATH) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-fPIC', '-shared', '-o', SO_PATH, src, '-lm'])
    return SO_PATH


def build_sanitized():
    src = os.path.join(_HERE, 'bprmf_oracle.c')
    if not os.path.isfile(SAN_SO_PATH) or os.path.getmtime(SAN_SO_PATH) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O1', '-g', '-fno-omit-frame-pointer', '-fsanitize=address,undefined',
                               '-fno-sanitize-recover=undefined', '-ffp-contract=off', '-fPIC', '-shared', '-o',
                               SAN_SO_PATH, src, '-lm'])
    return SAN_SO_PATH


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get('HSK_ORACLE_SO') or SO_PATH
        if path == SO_PATH and not os.path.isfile(SO_PATH):
            build()
        _lib = ctypes.CDLL(path)
        _lib.orc_bpr_loss_grad.restype = c_double
        _lib.orc_count_bad_negatives.restype = c_int64
    return _lib


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def mf_scores(U, I, Ib, Ub, gb, u_idx, i_idx):
    """algorithms/sgd_alg.py:148-179"""
    U, I, Ib, Ub, gb = _f32(U), _f32(I), _f32(Ib), _f32(Ub), _f32(gb)
    u_idx, i_idx = _i64(u_idx), _i64(i_idx)
    B, K = i_idx.shape
    out = np.empty((B, K), dtype=np.float32)
    lib().orc_mf_scores(_p(U), _p(I), _p(Ib), _p(Ub), _p(gb), c_int64(U.shape[1]), _p(u_idx), _p(i_idx),
                        c_int64(B), c_int64(K), _p(out))
    return out


def bpr_loss_grad(logits, need_grad=True):
    """train/rec_losses.py:68-88 (+ autograd)"""
    logits = _f32(logits)
    B, K = logits.shape
    grad = np.empty_like(logits) if need_grad else None
    loss = lib().orc_bpr_loss_grad(_p(logits), c_int64(B), c_int64(K), _p(grad))
    return float(loss), grad


def rec_loss_grad(kind, logits, log_adjust=0.0, need_grad=True):
    """kind in {'bpr','bce','sampled_softmax'} (train/rec_losses.py:27-139) -> (loss, grad wrt logits)"""
    if kind == 'bpr':
        return bpr_loss_grad(logits, need_grad)
    logits = _f32(logits)
    B, K = logits.shape
    grad = np.empty_like(logits) if need_grad else None
    L = lib()
    L.orc_bce_loss_grad.restype = c_double
    L.orc_ssm_loss_grad.restype = c_double
    if kind == 'bce':
        loss = L.orc_bce_loss_grad(_p(logits), c_int64(B), c_int64(K), _p(grad))
    elif kind == 'sampled_softmax':
        loss = L.orc_ssm_loss_grad(_p(logits), c_int64(B), c_int64(K), c_double(log_adjust), _p(grad))
    else:
        raise ValueError(kind)
    return float(loss), grad


def mf_backward(U, I, u_idx, i_idx, g, item_bias=True, user_bias=False, global_bias=False):
    """autograd backward of mf_scores -> dense grads"""
    U, I, g = _f32(U), _f32(I), _f32(g)
    u_idx, i_idx = _i64(u_idx), _i64(i_idx)
    B, K = i_idx.shape
    gU, gI = np.empty_like(U), np.empty_like(I)
    gIb = np.empty(I.shape[0], np.float32) if item_bias else None
    gUb = np.empty(U.shape[0], np.float32) if user_bias else None
    ggb = np.empty(1, np.float32) if global_bias else None
    lib().orc_mf_backward(_p(U), _p(I), c_int64(U.shape[0]), c_int64(I.shape[0]), c_int64(U.shape[1]), _p(u_idx),
                          _p(i_idx), c_int64(B), c_int64(K), _p(g), _p(gU), _p(gI), _p(gIb), _p(gUb), _p(ggb))
    return gU, gI, gIb, gUb, ggb


def adamw_step(p, g, m, v, lr, wd, step, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor step, in place on float32 numpy arrays (g None = zeros)."""
    for a in (p, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    g = _f32(g)
    lib().orc_adamw_step(_p(p), _p(g), _p(m), _p(v), c_int64(p.size), c_double(lr), c_double(b1), c_double(b2),
                         c_double(eps), c_double(wd), c_int64(step))


def opt_step(optimizer, p, g, m, v, lr, wd, step, b1=0.9, b2=0.999, eps=None):
    """One step of torch.optim.{AdamW, Adam, Adagrad} with the reference's settings (train/trainer.py:48-53), in place.
    v is exp_avg_sq or adagrad's state_sum; m is untouched by adagrad."""
    if optimizer == 'adamw':
        return adamw_step(p, g, m, v, lr, wd, step, b1, b2, 1e-8 if eps is None else eps)
    for a in (p, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    g = _f32(g)
    if optimizer == 'adam':
        lib().orc_adam_step(_p(p), _p(g), _p(m), _p(v), c_int64(p.size), c_double(lr), c_double(b1), c_double(b2),
                            c_double(1e-8 if eps is None else eps), c_double(wd), c_int64(step))
    elif optimizer == 'adagrad':
        lib().orc_adagrad_step(_p(p), _p(g), _p(v), c_int64(p.size), c_double(lr),
                               c_double(1e-10 if eps is None else eps), c_double(wd))
    else:
        raise ValueError(optimizer)


class MfOracleTrainer:
    """Dense restatement of one Trainer.fit step (train/trainer.py:128-148) for MF + BPR + AdamW.

    d loss/d user_bias and d loss/d global_bias are defined as exactly 0 (they cancel in
    s_pos - s_neg; the reference's ~1e-9 autograd noise there is amplified by Adam and is not part
    of the contract -- SURVEY.md section 7, hard part 2)."""

    def __init__(self, U, I, Ib=None, Ub=None, gb=None, lr=1e-3, wd=0.0, loss='bpr', log_adjust=0.0, optimizer='adamw'):
        self.loss, self.log_adjust, self.optimizer = loss, log_adjust, optimizer
        self.P = {'user_emb': _f32(U).copy(), 'item_emb': _f32(I).copy()}
        if Ib is not None:
            self.P['item_bias'] = _f32(Ib).reshape(-1).copy()
        if Ub is not None:
            self.P['user_bias'] = _f32(Ub).reshape(-1).copy()
        if gb is not None:
            self.P['global_bias'] = _f32(gb).reshape(-1).copy()
        self.M = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.V = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.lr, self.wd, self.t = lr, wd, 0

    def forward(self, u_idx, i_idx):
        P = self.P
        return mf_scores(P['user_emb'], P['item_emb'], P.get('item_bias'), P.get('user_bias'), P.get('global_bias'),
                         u_idx, i_idx)

    def step(self, u_idx, i_idx):
        P = self.P
        logits = self.forward(u_idx, i_idx)
        loss, g = rec_loss_grad(self.loss, logits, self.log_adjust)
        # bpr / sampled_softmax: the user and global bias cancel in the loss (zero gradient by definition);
        # bce sees every logit on its own, so they do receive gradient
        bce = self.loss == 'bce'
        gU, gI, gIb, gUb, ggb = mf_backward(P['user_emb'], P['item_emb'], u_idx, i_idx, g, item_bias='item_bias' in P,
                                            user_bias=bce and 'user_bias' in P, global_bias=bce and 'global_bias' in P)
        grads = {'user_emb': gU, 'item_emb': gI, 'item_bias': gIb, 'user_bias': gUb, 'global_bias': ggb}
        self.t += 1
        for k in P:
            opt_step(self.optimizer, P[k], grads[k], self.M[k], self.V[k], self.lr, self.wd, self.t)
        return loss, logits, g, grads


class BiasOracleTrainer:
    """SGDBaseline (algorithms/sgd_alg.py:72-107): logits = ub[u] + ib[i] + gb, trained like MfOracleTrainer.
    Gradients are summed in fp32 in the reference's order (index_add over the flattened batch)."""

    def __init__(self, Ub, Ib, gb, lr=1e-3, wd=0.0, loss='bce', log_adjust=0.0, optimizer='adamw'):
        self.P = {'user_bias': _f32(Ub).reshape(-1).copy(), 'item_bias': _f32(Ib).reshape(-1).copy(),
                  'global_bias': _f32(gb).reshape(-1).copy()}
        self.M = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.V = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.loss, self.log_adjust, self.optimizer, self.lr, self.wd, self.t = loss, log_adjust, optimizer, lr, wd, 0

    def forward(self, u_idx, i_idx):
        P = self.P
        return ((P['user_bias'][u_idx][:, None] + P['item_bias'][i_idx]) + P['global_bias'][0]).astype(np.float32)

    def step(self, u_idx, i_idx):
        logits = self.forward(u_idx, i_idx)
        loss, g = rec_loss_grad(self.loss, logits, self.log_adjust)
        grads = {k: np.zeros_like(v) for k, v in self.P.items()}
        np.add.at(grads['item_bias'], i_idx.reshape(-1), g.reshape(-1))
        np.add.at(grads['user_bias'], u_idx, g.sum(axis=1, dtype=np.float32))
        grads['global_bias'][0] = g.sum(dtype=np.float32)
        self.t += 1
        for k in self.P:
            opt_step(self.optimizer, self.P[k], grads[k], self.M[k], self.V[k], self.lr, self.wd, self.t)
        return loss, logits, g, grads


def count_bad_negatives(indptr, indices, n_items, u_idx, neg):
    indptr, indices, u_idx, neg = _i64(indptr), _i32(indices), _i64(u_idx), _i64(neg)
    B, N = neg.shape
    return int(lib().orc_count_bad_negatives(_p(indptr), _p(indices), c_int64(n_items), _p(u_idx), _p(neg),
                                             c_int64(B), c_int64(N)))


def sample_negatives_reference_style(rng: np.random.RandomState, indptr, indices, n_items, u_idx, n_neg):
    """Restatement of TrainDataLoader._neg_sampling_collate_fn (data/dataloader.py:110-124): draw for all
    open slots with randint, re-test every row against the user's CSR row, repeat until no hit."""
    B = len(u_idx)
    neg = np.empty((B, n_neg), dtype=np.int64)
    open_ = np.ones((B, n_neg), dtype=bool)
    todo = open_.sum()
    while todo:
        neg[open_] = rng.randint(0, high=n_items, size=todo)
        for b in range(B):
            u = u_idx[b]
            open_[b] = np.isin(neg[b], indices[indptr[u]:indptr[u + 1]])
        todo = open_.sum()
    return neg


def eval_scores(U, I, Ib, Ub, gb, u_idx, excl_indptr=None, excl_indices=None):
    """eval/eval.py:240-251"""
    U, I, Ib, Ub, gb = _f32(U), _f32(I), _f32(Ib), _f32(Ub), _f32(gb)
    u_idx = _i64(u_idx)
    R, n_items = len(u_idx), I.shape[0]
    out = np.empty((R, n_items), dtype=np.float32)
    ip = None if excl_indptr is None else _i64(excl_indptr)
    ii = None if excl_indices is None else _i32(excl_indices)
    lib().orc_eval_scores(_p(U), _p(I), _p(Ib), _p(Ub), _p(gb), c_int64(n_items), c_int64(U.shape[1]), _p(u_idx),
                          c_int64(R), _p(ip), _p(ii), _p(out))
    return out


def topk(x, k):
    """logits.topk(k): values desc, ties by lower index"""
    x = _f32(x)
    R, C = x.shape
    vals = np.empty((R, k), np.float32)
    idx = np.empty((R, k), np.int64)
    lib().orc_topk(_p(x), c_int64(R), c_int64(C), c_int64(k), _p(vals), _p(idx))
    return vals, idx


def rank_metrics(topk_idx, u_idx, indptr, indices, ks):
    """eval/metrics.py:4-105 -> [R, len(ks), 3] (precision, recall, ndcg)"""
    topk_idx, u_idx, indptr, indices = _i64(topk_idx), _i64(u_idx), _i64(indptr), _i32(indices)
    ks_a = _i32(ks)
    R, kmax = topk_idx.shape
    out = np.empty((R, len(ks), 3), np.float32)
    lib().orc_rank_metrics(_p(topk_idx), c_int64(R), c_int64(kmax), _p(u_idx), _p(indptr), _p(indices), _p(ks_a),
                           c_int32(len(ks)), _p(out))
    return out


def full_eval_metrics(U, I, Ib, Ub, gb, u_all, excl_indptr, excl_indices, lab_indptr, lab_indices,
                      ks=(5, 10, 50, 100), user_group=None, n_groups=0, batch=256):
    """evaluate_recommender_algorithm + FullEvaluator (eval/eval.py:54-118,237-255): mean metrics over
    users, plus `group_<g>_` variants."""
    kmax = max(ks)
    sums = {}
    counts = {-1: 0}
    for lo in range(0, len(u_all), batch):
        ub = np.asarray(u_all[lo:lo + batch], dtype=np.int64)
        sc = eval_scores(U, I, Ib, Ub, gb, ub, excl_indptr, excl_indices)
        _, ids = topk(sc, kmax)
        met = rank_metrics(ids, ub, lab_indptr, lab_indices, ks)
        counts[-1] += len(ub)
        groups = [(-1, np.ones(len(ub), bool))]
        if n_groups > 0:
            for g in range(n_groups):
                sel = np.asarray(user_group)[ub] == g
                counts[g] = counts.get(g, 0) + int(sel.sum())
                groups.append((g, sel))
        for g, sel in groups:
            for t, k in enumerate(ks):
                for j, name in enumerate(('precision', 'recall', 'ndcg')):
                    key = (g, f'{name}@{k}')
                    sums[key] = sums.get(key, 0.0) + float(met[sel, t, j].astype(np.float64).sum())
    out = {}
    for (g, name), s in sums.items():
        out[name if g == -1 else f'group_{g}_{name}'] = s / counts[g]
    return out


# ------------------------------------------------------------------------------------------------
# synthetic interactions generated on the device (hassaku_amd/csrc/hsk_synth.hip): numpy restatement of the law
# (no reference counterpart: the reference reads processed CSVs, data/dataset.py:120-131; this pins the generator
# the cfg5 workload trains on, so that a test can rebuild any user's row on the host)
# ------------------------------------------------------------------------------------------------
def _philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 on uint32 numpy arrays (the same rounds as hsk_philox4x32_10 in csrc/hsk_common.h)."""
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32) for x in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0.astype(np.uint64)
        p1 = M1 * c2.astype(np.uint64)
        n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ np.uint32(k0)
        n1 = p1.astype(np.uint32)
        n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ np.uint32(k1)
        n3 = p0.astype(np.uint32)
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def synth_degrees(users, deg_min, deg_span, seed):
    """deg(u) of hsk_synth_degrees for an array of user ids."""
    u = np.asarray(users, dtype=np.uint64)
    z = np.zeros(len(u), dtype=np.uint32)
    r = _philox4x32_10((u & np.uint64(0xFFFFFFFF)).astype(np.uint32), (u >> np.uint64(32)).astype(np.uint32), z,
                       z + np.uint32(0xD), seed & 0xFFFFFFFF, seed >> 32)[0]
    return deg_min + (r % np.uint32(deg_span)).astype(np.int64)


def synth_rows(users, n_items, deg_min, deg_span, skew, seed):
    """The item rows hsk_synth_fill writes for `users` -> list of int32 arrays (sorted, duplicate-free)."""
    users = np.asarray(users, dtype=np.int64)
    degs = synth_degrees(users, deg_min, deg_span, seed)
    out = []
    for u, deg in zip(users, degs):
        j = np.arange(deg)
        uu = np.full(deg, u, dtype=np.uint64)
        r = _philox4x32_10((uu & np.uint64(0xFFFFFFFF)).astype(np.uint32), (uu >> np.uint64(32)).astype(np.uint32),
                           (j >> 2).astype(np.uint32), np.full(deg, 0xE, dtype=np.uint32), seed & 0xFFFFFFFF, seed >> 32)
        w = np.choose(j & 3, r).astype(np.float64)
        x = (j.astype(np.float64) + (w + 0.5) * (1.0 / 4294967296.0)) / np.float64(deg)
        t = x.copy()
        for _ in range(1, skew):
            t = t * x
        v = np.minimum((t * np.float64(n_items)).astype(np.int64), n_items - 1)
        for q in range(1, deg):
            if v[q] <= v[q - 1]:
                v[q] = v[q - 1] + 1
        for q in range(deg - 1, -1, -1):
            cap = n_items - 1 if q == deg - 1 else v[q + 1] - 1
            v[q] = min(v[q], cap)
        out.append(v.astype(np.int32))
    return out
