"""Pins the CPU oracle (oracle/) against golden vectors produced by the reference itself
(oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import G1_CASES, PARAM_KEYS, assert_adam_param_close, csr_from_pairs, load_golden, max_norm_err

RTOL = 1e-5  # BASELINE.json north_star: within 1e-5 relative fp32


def _init_params(fx):
    P = {}
    for sk, name in PARAM_KEYS.items():
        if 'init.' + sk in fx:
            P[name] = fx['init.' + sk]
    return P


@pytest.mark.parametrize('case', G1_CASES)
def test_forward_loss_grads_match_reference(oracle, case):
    fx = load_golden(f'g1_step_{case}.npz')
    P = _init_params(fx)
    u, i = fx['s1.u_idx'], fx['s1.i_idx']
    logits = oracle.mf_scores(P['user_emb'], P['item_emb'], P.get('item_bias'), P.get('user_bias'),
                              P.get('global_bias'), u, i)
    np.testing.assert_allclose(logits, fx['s1.logits'], rtol=RTOL, atol=1e-9)
    loss, g = oracle.bpr_loss_grad(fx['s1.logits'])
    assert abs(loss - float(fx["s1.loss"])) <= 1e-7 * abs(float(fx["s1.loss"]))
    np.testing.assert_allclose(g, fx['s1.grad_logits'], rtol=RTOL, atol=1e-12)
    gU, gI, gIb, gUb, ggb = oracle.mf_backward(P['user_emb'], P['item_emb'], u, i, fx['s1.grad_logits'],
                                               item_bias=True, user_bias=True, global_bias=True)
    assert max_norm_err(gU, fx['s1.grad.user_embeddings.weight']) < RTOL
    assert max_norm_err(gI, fx['s1.grad.item_embeddings.weight']) < RTOL
    if 'item_bias' in P:
        assert max_norm_err(gIb, fx['s1.grad.item_bias.weight'].reshape(-1)) < RTOL
    # user/global bias gradients are mathematically zero; the reference holds rounding noise there
    if 'user_bias' in P:
        scale = np.abs(fx['s1.grad_logits']).max()
        assert np.abs(gUb).max() < 1e-5 * scale and np.abs(fx['s1.grad.user_bias.weight']).max() < 1e-5 * scale
        assert abs(float(ggb[0])) < 1e-5 * scale


# Adam divides by sqrt(v)+eps: for gradient elements with |g| <~ eps (1e-8) a last-bit difference in the
# fp32 gradient SUM (summation order) moves the parameter by up to ~lr*dg/(4*eps).  The optimiser arithmetic
# itself is pinned to 1e-6 below (test_adamw_on_reference_grads); end-to-end tensors get this looser bound.


def _diffs(logits):
    return logits[:, :1] - logits[:, 1:]


@pytest.mark.parametrize('case', G1_CASES)
def test_three_adamw_steps_match_reference(oracle, case):
    fx = load_golden(f'g1_step_{case}.npz')
    P = _init_params(fx)
    tr = oracle.MfOracleTrainer(P['user_emb'], P['item_emb'], P.get('item_bias'), P.get('user_bias'),
                                P.get('global_bias'), lr=float(fx['lr']), wd=float(fx['wd']))
    noisy_bias = 'user_bias' in P or 'global_bias' in P
    for step in (1, 2, 3):
        loss, logits, _, _ = tr.step(fx[f's{step}.u_idx'], fx[f's{step}.i_idx'])
        if noisy_bias and step > 1:
            # user/global bias follow amplified noise in the reference; they cancel in pos - neg
            np.testing.assert_allclose(_diffs(logits), _diffs(fx[f's{step}.logits']), rtol=1e-3, atol=2e-6)
        else:
            np.testing.assert_allclose(logits, fx[f's{step}.logits'], rtol=1e-3, atol=2e-6)
        assert abs(loss - float(fx[f's{step}.loss'])) <= 1e-6 * abs(float(fx[f's{step}.loss']))
        if step in (1, 3):
            for sk, name in PARAM_KEYS.items():
                if name in ('user_bias', 'global_bias') or name not in tr.P:
                    continue  # zero-gradient parameters: reference = amplified noise (hard part 2)
                ref = fx[f's{step}.param.{sk}'].reshape(tr.P[name].shape)
                assert_adam_param_close(tr.P[name], ref, (step, name))
                if step == 1:
                    assert max_norm_err(tr.M[name], fx[f's{step}.m.{sk}'].reshape(ref.shape)) < RTOL, (step, name)
                    assert max_norm_err(tr.V[name], fx[f's{step}.v.{sk}'].reshape(ref.shape)) < RTOL, (step, name)
                else:
                    assert_adam_param_close(tr.M[name], fx[f's{step}.m.{sk}'], (step, 'm', name))
                    assert_adam_param_close(tr.V[name], fx[f's{step}.v.{sk}'], (step, 'v', name))


@pytest.mark.parametrize('case', G1_CASES)
def test_adamw_on_reference_grads(oracle, case):
    """The optimiser alone, fed the reference's own dense gradients: p, exp_avg, exp_avg_sq after step 1."""
    fx = load_golden(f'g1_step_{case}.npz')
    for sk in PARAM_KEYS:
        if 'init.' + sk not in fx:
            continue
        p = fx['init.' + sk].copy()
        m, v = np.zeros_like(p), np.zeros_like(p)
        oracle.adamw_step(p, fx['s1.grad.' + sk], m, v, float(fx['lr']), float(fx['wd']), 1)
        assert max_norm_err(p, fx['s1.param.' + sk]) < 1e-6, sk
        assert max_norm_err(m, fx['s1.m.' + sk]) < 1e-6, sk
        assert max_norm_err(v, fx['s1.v.' + sk]) < 1e-6, sk


def test_adamw_zero_grad_rows_keep_moving(oracle):
    """Dense AdamW semantics (hard part 1): rows outside the batch still decay and move."""
    fx = load_golden('g1_step_d64_item.npz')
    u1 = set(fx['s1.u_idx'].tolist())
    u2 = set(fx['s2.u_idx'].tolist()) | set(fx['s3.u_idx'].tolist())
    only_first = sorted(u1 - u2)
    assert only_first, 'fixture should contain a user touched in step 1 only'
    r = only_first[0]
    p1 = fx['s1.param.user_embeddings.weight'][r]
    p3 = fx['s3.param.user_embeddings.weight'][r]
    assert np.abs(p3 - p1).max() > 1e-5  # kept moving on zero gradients (momentum + decay)


@pytest.mark.parametrize('split', ['val', 'test'])
def test_eval_scores_topk_metrics_match_reference(oracle, split):
    fx = load_golden('g3_eval.npz')
    U, I = fx['param.user_embeddings.weight'], fx['param.item_embeddings.weight']
    Ib, Ub, gb = fx['param.item_bias.weight'].reshape(-1), fx['param.user_bias.weight'].reshape(-1), fx['param.global_bias']
    n_users = int(fx['n_users'])
    excl = fx['train'] if split == 'val' else np.concatenate([fx['train'], fx['val']])
    e_ptr, e_idx = csr_from_pairs(excl, n_users)
    l_ptr, l_idx = csr_from_pairs(fx[split], n_users)
    u = fx[f'{split}.u']
    sc = oracle.eval_scores(U, I, Ib, Ub, gb, u, e_ptr, e_idx)
    ref = fx[f'{split}.masked_logits']
    assert np.array_equal(np.isinf(sc), np.isinf(ref))
    fin = ~np.isinf(ref)
    np.testing.assert_allclose(sc[fin], ref[fin], rtol=RTOL, atol=2e-6)  # |scores| <= ~6, fp32 sums cancel
    # top-100 ids: identical wherever the reference's scores are not tied within rounding
    _, ids = oracle.topk(ref, 100)
    assert np.array_equal(ids, fx[f'{split}.top100'])
    m = oracle.full_eval_metrics(U, I, Ib, Ub, gb, np.arange(n_users), e_ptr, e_idx, l_ptr, l_idx,
                                 user_group=fx['user_group'], n_groups=2, batch=16)
    names = [str(x) for x in fx[f'{split}.metric_names']]
    assert sorted(m) == names
    for name, val in zip(names, fx[f'{split}.metric_values']):
        assert abs(m[name] - val) <= 1e-6 + 1e-5 * abs(val), name


def test_eval_d512_wide_catalogue_matches_reference(oracle):
    """G3 at D = 512 over 4224 items (the BASELINE embedding size, a catalogue wide enough for the wide-row top-k)."""
    from conftest import g3_d512_params
    fx = load_golden('g3_eval_d512.npz')
    U, I, Ib = g3_d512_params(fx)
    n_users = int(fx['n_users'])
    e_ptr, e_idx = csr_from_pairs(fx['train'], n_users)
    l_ptr, l_idx = csr_from_pairs(fx['val'], n_users)
    sc = oracle.eval_scores(U, I, Ib, None, None, fx['val.u'], e_ptr, e_idx)
    ref = fx['val.masked_logits']
    assert np.array_equal(np.isinf(sc), np.isinf(ref))
    fin = ~np.isinf(ref)
    np.testing.assert_allclose(sc[fin], ref[fin], rtol=RTOL, atol=2e-6 * np.abs(ref[fin]).max())
    _, ids = oracle.topk(ref, 100)
    assert np.array_equal(ids, fx['val.top100'])
    m = oracle.full_eval_metrics(U, I, Ib, None, None, np.arange(n_users), e_ptr, e_idx, l_ptr, l_idx,
                                 user_group=fx['user_group'], n_groups=2, batch=16)
    for name, val in zip([str(x) for x in fx['val.metric_names']], fx['val.metric_values']):
        assert abs(m[name] - val) <= 1e-6 + 1e-5 * abs(val), name


def test_replay_of_reference_fit_matches(oracle):
    """G4: replay the exact batch stream the reference loader produced through the oracle trainer."""
    fx = load_golden('g4_fit.npz')
    tr = oracle.MfOracleTrainer(fx['init.user_embeddings.weight'], fx['init.item_embeddings.weight'],
                                fx['init.item_bias.weight'], lr=float(fx['lr']), wd=float(fx['wd']))
    for s in range(int(fx['n_steps'])):
        tr.step(fx[f'b{s}.u'], fx[f'b{s}.i'])
    assert max_norm_err(tr.P['user_emb'], fx['final.user_embeddings.weight']) < RTOL
    assert max_norm_err(tr.P['item_emb'], fx['final.item_embeddings.weight']) < RTOL
    assert max_norm_err(tr.P['item_bias'], fx['final.item_bias.weight'].reshape(-1)) < RTOL
    n_users = int(fx['n_users'])
    e_ptr, e_idx = csr_from_pairs(fx['train'], n_users)
    l_ptr, l_idx = csr_from_pairs(fx['val'], n_users)
    m = oracle.full_eval_metrics(tr.P['user_emb'], tr.P['item_emb'], tr.P['item_bias'], None, None,
                                 np.arange(n_users), e_ptr, e_idx, l_ptr, l_idx, batch=16)
    names = [str(x) for x in fx['val_metric_names']]
    for name, val in zip(names, fx['val_metric_values_last']):
        assert abs(m[name] - val) <= 1e-6 + 1e-5 * abs(val), name


def test_reference_batches_respect_sampler_invariants(oracle):
    """The batches the reference loader emitted: negatives in range and never a train positive."""
    fx = load_golden('g4_fit.npz')
    n_users, n_items = int(fx['n_users']), int(fx['n_items'])
    ptr, idx = csr_from_pairs(fx['train'], n_users)
    for s in range(int(fx['n_steps'])):
        u, i = fx[f'b{s}.u'], fx[f'b{s}.i']
        assert oracle.count_bad_negatives(ptr, idx, n_items, u, i[:, 1:]) == 0
        # column 0 is the positive: it IS in the user's row
        assert oracle.count_bad_negatives(ptr, idx, n_items, u, i[:, :1]) == len(u)


def test_rank_metrics_match_reference_functions(oracle):
    fx = load_golden('g5_metrics.npz')
    y = fx['y_true']
    R = y.shape[0]
    pairs = np.argwhere(y > 0)
    ptr, idx = csr_from_pairs(pairs, R)
    _, ids = oracle.topk(fx['logits'], 100)
    assert np.array_equal(ids, fx['top100'])
    ks = [5, 10, 50, 100]
    m = oracle.rank_metrics(ids, np.arange(R), ptr, idx, ks)
    for t, k in enumerate(ks):
        for j, name in enumerate(('precision', 'recall', 'ndcg')):
            np.testing.assert_allclose(m[:, t, j], fx[f'{name}@{k}'], rtol=1e-5, atol=1e-7, err_msg=f'{name}@{k}')


def test_metric_known_answers(oracle):
    """Closed-form cases of the reference's only unit test (framework_tests/eval/test_metrics.py:10-69):
    10 users x 20 items, logits strictly decreasing in the item index, k = 10."""
    import math
    R, I, k = 10, 20, 10
    logits = np.tile(np.arange(I, 0, -1, dtype=np.float32), (R, 1))
    _, ids = oracle.topk(logits, k)
    assert np.array_equal(ids[0], np.arange(k))

    def metrics(cols):
        pairs = [(r, c) for r in range(R) for c in cols]
        ptr, idx = csr_from_pairs(np.array(pairs).reshape(-1, 2), R)
        return oracle.rank_metrics(ids, np.arange(R), ptr, idx, [k]).mean(axis=0)[0]

    disc = [1.0 / math.log2(j + 2) for j in range(k)]
    p, r, n = metrics([])
    assert (p, r, n) == (0, 0, 0)
    p, r, n = metrics(range(I))
    assert p == 1 and abs(r - k / I) < 1e-6 and abs(n - 1) < 1e-6
    p, r, n = metrics([0])
    assert abs(p - 1 / k) < 1e-6 and r == 1 and n == 1
    p, r, n = metrics([1, 2])
    assert abs(p - 2 / k) < 1e-6 and r == 1
    assert abs(n - (math.log2(4) + math.log2(3)) / (math.log2(4) * (1 + math.log2(3)))) < 1e-5
    p, r, n = metrics([0] + list(range(k + 1, I)))
    assert abs(p - 1 / k) < 1e-6 and abs(r - 1 / (I - k)) < 1e-6
    assert abs(n - 1 / sum(disc[:min(k, I - k)])) < 1e-5
