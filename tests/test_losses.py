"""bce and sampled_softmax: oracle pinned to the reference's golden vectors (CPU), then the HIP
un-fused operator, the autograd path and the fused step held to the same vectors (GPU)."""
import numpy as np
import pytest
import torch

from conftest import PARAM_KEYS, assert_adam_param_close, load_golden, max_norm_err

G6_CASES = [('bce_d32_item', 'bce'), ('bce_d32_all', 'bce'), ('ssm_d32_item', 'sampled_softmax'),
            ('ssm_d402_all', 'sampled_softmax')]
RTOL = 1e-5


def _init(fx):
    return {name: fx['init.' + sk] for sk, name in PARAM_KEYS.items() if 'init.' + sk in fx}


def _ref_grad_logits(fx, kind):
    """d loss / d logits as the reference's autograd sees it.  For sampled_softmax the fixture's retained gradient
    belongs to the logits AFTER the in-place `logits[:, 1:] += log(I/N)` (train/rec_losses.py:134), i.e. it lacks
    the -1/B that flows into column 0 through `-logits[:, 0]`, taken before that in-place op; add it back."""
    g = fx['s1.grad_logits'].copy()
    if kind == 'sampled_softmax':
        g[:, 0] -= 1.0 / g.shape[0]
    return g


def _zero_grad_biases(kind):
    """parameters whose gradient is identically zero under the loss (reference holds amplified noise there)"""
    return () if kind == 'bce' else ('user_bias', 'global_bias')


@pytest.mark.parametrize('case,kind', G6_CASES)
def test_oracle_loss_and_grads_match_reference(oracle, case, kind):
    fx = load_golden(f'g6_{case}.npz')
    adj = float(fx['log_adjust'])
    loss, g = oracle.rec_loss_grad(kind, fx['s1.logits'], adj)
    assert abs(loss - float(fx['s1.loss'])) <= 2e-6 * abs(float(fx['s1.loss']))   # ssm loss is fp32 in the reference
    np.testing.assert_allclose(g, _ref_grad_logits(fx, kind), rtol=2e-5, atol=1e-9)
    P = _init(fx)
    gU, gI, gIb, gUb, ggb = oracle.mf_backward(P['user_emb'], P['item_emb'], fx['s1.u_idx'], fx['s1.i_idx'],
                                               _ref_grad_logits(fx, kind), True, True, True)
    assert max_norm_err(gU, fx['s1.grad.user_embeddings.weight']) < RTOL
    assert max_norm_err(gI, fx['s1.grad.item_embeddings.weight']) < RTOL
    if kind == 'bce' and 'user_bias' in P:   # bce does send gradient to the user / global bias
        assert max_norm_err(gUb, fx['s1.grad.user_bias.weight'].reshape(-1)) < RTOL
        assert abs(float(ggb[0]) - float(fx['s1.grad.global_bias'][0])) < RTOL * abs(float(fx['s1.grad.global_bias'][0]))


@pytest.mark.parametrize('case,kind', G6_CASES)
def test_oracle_three_steps_match_reference(oracle, case, kind):
    fx = load_golden(f'g6_{case}.npz')
    P = _init(fx)
    tr = oracle.MfOracleTrainer(P['user_emb'], P['item_emb'], P.get('item_bias'), P.get('user_bias'), P.get('global_bias'),
                                lr=float(fx['lr']), wd=float(fx['wd']), loss=kind, log_adjust=float(fx['log_adjust']))
    for step in (1, 2, 3):
        loss, _, _, _ = tr.step(fx[f's{step}.u_idx'], fx[f's{step}.i_idx'])
        assert abs(loss - float(fx[f's{step}.loss'])) <= 5e-6 * abs(float(fx[f's{step}.loss'])), step
    for sk, name in PARAM_KEYS.items():
        if name not in tr.P or name in _zero_grad_biases(kind):
            continue
        assert_adam_param_close(tr.P[name], fx[f's3.param.{sk}'], name)


def test_loss_registry_builds_all_three():
    from hassaku_amd.train.rec_losses import RecommenderSystemLossesEnum
    class DS:
        n_items = 1000
    conf = {'train_neg_strategy': 'uniform', 'neg_train': 10}
    bce = RecommenderSystemLossesEnum['bce'].value.build_from_conf(conf, DS)
    ssm = RecommenderSystemLossesEnum['sampled_softmax'].value.build_from_conf(conf, DS)
    assert bce.name == 'RecBinaryCrossEntropy' and ssm.name == 'RecSampledSoftmaxLoss'
    assert abs(ssm.log_adjust - np.log(100.0)) < 1e-12
    conf['train_neg_strategy'] = 'popular'
    assert RecommenderSystemLossesEnum['sampled_softmax'].value.build_from_conf(conf, DS).log_adjust == 0.0


# ---------------------------------------------------------------------------------------------------
def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize('case,kind', G6_CASES)
def test_hip_unfused_loss_vs_golden(case, kind):
    from hassaku_amd import hip_ops as ops
    fx = load_golden(f'g6_{case}.npz')
    loss, g = ops.rec_loss_grad(kind, dev(fx['s1.logits']), float(fx['log_adjust']))
    assert abs(loss.item() - float(fx['s1.loss'])) <= 2e-6 * abs(float(fx['s1.loss']))
    np.testing.assert_allclose(g.cpu().numpy(), _ref_grad_logits(fx, kind), rtol=2e-5, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize('case,kind', G6_CASES)
def test_hip_autograd_path_vs_golden(case, kind):
    """model(u, i) -> rec_loss.compute_loss -> backward: dense grads of every parameter, incl. the biases under bce."""
    from hassaku_amd.algorithms.sgd_alg import SGDMatrixFactorization
    from hassaku_amd.train.rec_losses import RecBinaryCrossEntropy, RecSampledSoftmaxLoss
    fx = load_golden(f'g6_{case}.npz')
    model = SGDMatrixFactorization(int(fx['n_users']), int(fx['n_items']), int(fx['dim']), bool(fx['use_user_bias']),
                                   bool(fx['use_item_bias']), bool(fx['use_global_bias']))
    model.load_state_dict({k[5:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith('init.')})
    model.to('cuda')
    loss_fn = RecBinaryCrossEntropy() if kind == 'bce' else RecSampledSoftmaxLoss(int(fx['n_items']), 'uniform', int(fx['n_neg']))
    out = model(dev(fx['s1.u_idx']), dev(fx['s1.i_idx']))
    np.testing.assert_allclose(out.detach().cpu().numpy(), fx['s1.logits'], rtol=RTOL, atol=1e-7)
    loss = loss_fn.compute_loss(out, None)
    assert abs(loss.item() - float(fx['s1.loss'])) <= 2e-6 * abs(float(fx['s1.loss']))
    loss.backward()
    for name, p in model.named_parameters():
        ref = fx['s1.grad.' + name]
        if name in ('user_bias.weight', 'global_bias') and kind != 'bce':
            assert p.grad.abs().max().item() < 1e-5 * np.abs(fx['s1.grad_logits']).max()
        else:
            assert max_norm_err(p.grad.cpu().numpy(), ref) < RTOL, name


@pytest.mark.gpu
@pytest.mark.parametrize('case,kind', [c for c in G6_CASES if c[0] != 'bce_d32_all'])
def test_hip_fused_step_vs_golden(case, kind):
    from hassaku_amd import hip_ops as ops
    fx = load_golden(f'g6_{case}.npz')
    P = _init(fx)
    t = {k: dev(v.reshape(-1) if k not in ('user_emb', 'item_emb') else v) for k, v in P.items()}
    B, K = fx['s1.i_idx'].shape
    st = ops.BprMfFusedState(t['user_emb'], t['item_emb'], t.get('item_bias'), t.get('user_bias'), t.get('global_bias'),
                             lr=float(fx['lr']), wd=float(fx['wd']), max_batch=B, max_cols=K, loss=kind,
                             log_adjust=float(fx['log_adjust']))
    for step in (1, 2, 3):
        st.step(dev(fx[f's{step}.u_idx']), dev(fx[f's{step}.i_idx']))
        assert abs(st.last_loss() - float(fx[f's{step}.loss'])) <= 5e-6 * abs(float(fx[f's{step}.loss'])), step
    st.flush()
    for sk, name in PARAM_KEYS.items():
        if name not in P or name in _zero_grad_biases(kind):
            continue
        assert_adam_param_close(t[name].cpu().numpy().reshape(-1), fx[f's3.param.{sk}'].reshape(-1), name)
    st.check_status()


@pytest.mark.gpu
def test_fused_bce_refuses_user_or_global_bias():
    from hassaku_amd import hip_ops as ops
    with pytest.raises(ValueError):
        ops.BprMfFusedState(torch.zeros(4, 8, device='cuda'), torch.zeros(5, 8, device='cuda'), None,
                            torch.zeros(4, device='cuda'), None, lr=1e-3, wd=0., max_batch=2, max_cols=3, loss='bce')


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['bce', 'sampled_softmax'])
def test_fused_losses_vs_oracle_at_baseline_shape(oracle, kind):
    """D=512, N=100, B=256 (the cfg3 row shape): two steps of the fused kernel against the oracle."""
    from hassaku_amd import hip_ops as ops
    rng = np.random.RandomState(5)
    U, I, D, B, N = 300, 500, 512, 256, 100
    P = {'user_emb': (rng.randn(U, D) * 0.08).astype(np.float32), 'item_emb': (rng.randn(I, D) * 0.08).astype(np.float32),
         'item_bias': (rng.randn(I) * 0.1).astype(np.float32)}
    adj = float(np.log(I / N)) if kind == 'sampled_softmax' else 0.0
    t = {k: dev(v) for k, v in P.items()}
    st = ops.BprMfFusedState(t['user_emb'], t['item_emb'], t['item_bias'], lr=3e-4, wd=4e-5, max_batch=B, max_cols=N + 1,
                             loss=kind, log_adjust=adj)
    tr = oracle.MfOracleTrainer(P['user_emb'], P['item_emb'], P['item_bias'], lr=3e-4, wd=4e-5, loss=kind, log_adjust=adj)
    for _ in range(2):
        u = rng.randint(0, U, size=B).astype(np.int64)
        i = rng.randint(0, I, size=(B, N + 1)).astype(np.int64)
        st.step(dev(u), dev(i))
        ref, _, _, _ = tr.step(u, i)
        assert abs(st.last_loss() - ref) <= 2e-6 * abs(ref)
    st.flush()
    for name in P:
        assert_adam_param_close(t[name].cpu().numpy(), tr.P[name], name)


@pytest.mark.gpu
@pytest.mark.parametrize('kind,optimizer', [('bpr', 'adamw'), ('bce', 'adamw'), ('bpr', 'adagrad')])
def test_item_partitioned_forward_vs_oracle(oracle, kind, optimizer):
    """A shape that selects the item-partitioned forward (csrc/hsk_fwd_part.h: B = 2048, D = 256, a 6.1 MB item table ->
    P = 2) on loader-supplied batches with repeated users: three steps (the second and third find lazily updated rows
    behind, i.e. every unit replays them) against the CPU oracle -- loss to 2e-6, parameters by the Adam rule."""
    from hassaku_amd import hip_ops as ops
    rng = np.random.RandomState(9)
    U, I, D, B, N = 40000, 6000, 256, 2048, 16
    P = {'user_emb': (rng.randn(U, D) * 0.08).astype(np.float32), 'item_emb': (rng.randn(I, D) * 0.08).astype(np.float32),
         'item_bias': (rng.randn(I) * 0.1).astype(np.float32)}
    t = {k: dev(v) for k, v in P.items()}
    st = ops.BprMfFusedState(t['user_emb'], t['item_emb'], t['item_bias'], lr=3e-4, wd=4e-5, max_batch=B, max_cols=N + 1,
                             loss=kind, optimizer=optimizer)
    assert st.batch_columns(B, N + 1) == N + 2, 'this shape is meant to run the partitioned forward with P = 2'
    tr = oracle.MfOracleTrainer(P['user_emb'], P['item_emb'], P['item_bias'], lr=3e-4, wd=4e-5, loss=kind,
                                optimizer=optimizer)
    for _ in range(3):
        u = rng.randint(0, U, size=B).astype(np.int64)
        u[rng.rand(B) < 0.05] = u[0]                      # one user many times, others twice now and then
        i = rng.randint(0, I, size=(B, N + 1)).astype(np.int64)
        st.step(dev(u), dev(i))
        ref, _, _, _ = tr.step(u, i)
        assert abs(st.last_loss() - ref) <= 2e-6 * abs(ref)
    st.flush()
    st.check_status()
    tol = dict(max_tol=5e-3, frac=5e-3) if optimizer == 'adagrad' else {}
    for name in P:
        assert_adam_param_close(t[name].cpu().numpy(), tr.P[name], name, **tol)
