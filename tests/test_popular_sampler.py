"""train_neg_strategy 'popular' (SURVEY 8f, rank 2): negatives ~ pop^alpha restricted to the items the user has not
interacted with (NegativeSampler._neg_sample_popular + the rejection loop, data/dataloader.py:59-64,114-124)."""
import numpy as np
import pytest
import torch

from conftest import csr_from_pairs


def test_alias_table_reproduces_the_distribution():
    from hassaku_amd.hip_ops import build_alias_table
    rng = np.random.RandomState(0)
    for p in (rng.rand(257) ** 4, np.array([0.0, 1.0, 0.0, 3.0]), np.ones(5)):
        p = p / p.sum()
        prob, alias = build_alias_table(p)
        n = len(p)
        q = np.zeros(n)
        np.add.at(q, np.arange(n), prob / n)
        np.add.at(q, alias, (1.0 - prob) / n)
        assert np.abs(q - p).max() < 1e-7
    with pytest.raises(ValueError):
        build_alias_table(np.array([0.5, -0.1]))


def test_negative_sampler_accepts_popular(tmp_path):
    from hassaku_amd.data.dataloader import NegativeSampler
    from hassaku_amd.data.dataset import TrainRecDataset
    from hassaku_amd.data.synthetic import generate, write_csv_dataset
    write_csv_dataset(generate(30, 120, 600, seed=1), str(tmp_path))
    ds = TrainRecDataset(str(tmp_path))
    assert NegativeSampler(ds, 3, 'uniform').alias('cpu') is None
    s = NegativeSampler(ds, 3, 'popular', squashing_factor_pop_sampling=0.5)
    prob, idx = s.alias('cpu')
    assert prob.dtype == torch.float32 and idx.dtype == torch.int32 and prob.shape == (120,)
    p = np.power(ds.pop_distribution, 0.5)
    p /= p.sum()
    q = np.zeros(120)
    np.add.at(q, np.arange(120), prob.numpy() / 120)
    np.add.at(q, idx.numpy(), (1 - prob.numpy()) / 120)
    assert np.abs(q - p).max() < 1e-6


def _dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t.to(dt) if dt else t).cuda()


@pytest.mark.gpu
def test_popular_sampler_law_and_invariants(oracle):
    from hassaku_amd import hip_ops as ops
    rng = np.random.RandomState(2)
    n_users, n_items, n_neg = 12, 60, 30000
    pairs = np.argwhere(rng.rand(n_users, n_items) < 0.3)
    ptr, idx = csr_from_pairs(pairs, n_users)
    p = rng.rand(n_items) ** 3 + 1e-3
    p /= p.sum()
    prob, alias = ops.build_alias_table(p)
    u = np.arange(n_users, dtype=np.int64)
    status = ops.new_status('cuda')
    neg = ops.sample_negatives_uniform(_dev(ptr), _dev(idx), n_items, _dev(u), n_neg, seed=5, stream_id=1, status=status,
                                       alias=(_dev(prob), _dev(alias))).cpu().numpy()
    ops.raise_on_status(status)
    assert oracle.count_bad_negatives(ptr, idx, n_items, u, neg) == 0
    for b in (0, 5, 11):
        allowed = np.setdiff1d(np.arange(n_items), idx[ptr[b]:ptr[b + 1]])
        expect = p[allowed] / p[allowed].sum() * n_neg          # p restricted to the complement, renormalised
        cnt = np.bincount(neg[b], minlength=n_items)[allowed]
        chi2 = ((cnt - expect) ** 2 / expect).sum()
        assert chi2 < 2.2 * len(allowed), (b, chi2, len(allowed))
    # clearly NOT uniform: the most probable allowed item is drawn far more often than the least probable one
    allowed = np.setdiff1d(np.arange(n_items), idx[ptr[0]:ptr[1]])
    cnt0 = np.bincount(neg[0], minlength=n_items)
    hi, lo = allowed[np.argmax(p[allowed])], allowed[np.argmin(p[allowed])]
    assert cnt0[hi] > 5 * max(1, cnt0[lo])


@pytest.mark.gpu
def test_fused_step_samples_from_the_alias_table(oracle):
    from hassaku_amd import hip_ops as ops
    rng = np.random.RandomState(3)
    U, I, D, B, N = 64, 200, 32, 64, 100
    pairs = np.argwhere(rng.rand(U, I) < 0.05)
    ptr, idx = csr_from_pairs(pairs, U)
    p = np.zeros(I)
    p[:20] = 1.0                                     # only the first 20 items can ever be drawn
    prob, alias = ops.build_alias_table(p / p.sum())
    t = {'user_emb': torch.randn(U, D, device='cuda') * 0.05, 'item_emb': torch.randn(I, D, device='cuda') * 0.05,
         'item_bias': torch.zeros(I, device='cuda')}
    st = ops.BprMfFusedState(t['user_emb'], t['item_emb'], t['item_bias'], lr=1e-3, wd=0., max_batch=B, max_cols=N + 1,
                             seed=9, csr_indptr=_dev(ptr), csr_indices=_dev(idx), coo_user=_dev(pairs[:, 0], torch.int32),
                             coo_item=_dev(pairs[:, 1], torch.int32), alias=(_dev(prob), _dev(alias)))
    st.step_sampled(None, 0, B, N)
    u, i = st.last_batch(B, N + 1)
    u, i = u.cpu().numpy(), i.cpu().numpy()
    assert i[:, 1:].max() < 20
    assert oracle.count_bad_negatives(ptr, idx, I, u, i[:, 1:]) == 0
    st.check_status()
