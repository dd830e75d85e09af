"""BASELINE configs[4] (synthetic 100 M users x 10 M items, dim = 1024, neg_train = 200, 8 x MI355X row-parallel
embeddings): the on-device interaction generator, the shard-direct initialisation, and ONE RANK'S FULL-SIZE SHARE of the
eight -- 12.5 M x 1024 user rows + 1.25 M x 1024 item rows with their AdamW moments (169 GB, the first tables above 2^31
elements), lazy AdamW on both, the item range rank 3 of 8 owns of the 10 M-item catalogue -- stepped through the real
kernels and checked row by row against the oracle's arithmetic on the rows the steps touch.

Reference mechanisms replaced: nn.DataParallel (train/trainer.py:38-41), TrainRecDataset._prepare_data
(data/dataset.py:120-131), torch.optim.AdamW's dense update (train/trainer.py:52-53,147-148).
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

from test_dist import _free_port, _init  # noqa: E402


# ---------------------------------------------------------------------------------------------------
# CPU: the generator's law (numpy restatement) and the host-side sizing rules
# ---------------------------------------------------------------------------------------------------
def test_synthetic_law_rows_are_sorted_distinct_in_range():
    from oracle import oracle as orc
    users = np.concatenate([np.arange(300), [99_999_999, 2 ** 31 - 2]])
    for n_items, skew in ((10_000_000, 2), (1000, 1), (40, 3)):
        deg = orc.synth_degrees(users, 12, 17, seed=9)
        assert deg.min() >= 12 and deg.max() <= 28
        rows = orc.synth_rows(users, n_items, 12, 17, skew, seed=9)
        for r, d in zip(rows, deg):
            assert len(r) == d and r.dtype == np.int32
            assert r.min() >= 0 and r.max() < n_items and (np.diff(r) > 0).all()
    # mean degree 20, and a popularity skew towards low ids at skew = 2
    assert abs(orc.synth_degrees(np.arange(20000), 12, 17, seed=1).mean() - 20.0) < 0.2
    allr = np.concatenate(orc.synth_rows(np.arange(400), 1_000_000, 12, 17, 2, seed=1))
    assert (allr < 250_000).mean() > 0.45          # P(item < I/4) = sqrt(1/4) = 0.5 under x^2


def test_flush_cadence_rule_follows_table_and_batch():
    import ctypes
    from hassaku_amd import _lib
    lib = _lib.load()
    st = _lib.HskBprmfState()
    st.lazy_users, st.lazy_items, st.flush_every = 1, 1, 0
    NEVER = 1 << 30
    st.n_users, st.n_items, st.dim = 69878, 10677, 512          # ml10m: a user is in every 17th batch -> never sweep
    assert lib.hsk_bprmf_flush_cadence(ctypes.byref(st), 0, 4096) == NEVER
    st.n_users, st.n_items, st.dim = 12_500_000, 1_250_000, 1024   # a cfg5 shard: ~200 steps between the 300 GB sweeps
    f = lib.hsk_bprmf_flush_cadence(ctypes.byref(st), 0, 8750)
    assert 120 <= f <= 400, f
    assert lib.hsk_bprmf_flush_cadence(ctypes.byref(st), 1, 900_000) == NEVER   # 3 of 4 item rows touched every step
    st.flush_every = 64                                           # explicit cadence wins
    assert lib.hsk_bprmf_flush_cadence(ctypes.byref(st), 0, 8750) == 64
    st.lazy_users = 0
    assert lib.hsk_bprmf_flush_cadence(ctypes.byref(st), 0, 8750) == NEVER
    # the sharded carving leaves out the [max_batch, D] row buffers
    full = lib.hsk_bprmf_workspace_bytes(12_500_000, 1_250_000, 1024, 65536, 201)
    lean = lib.hsk_shard_base_workspace_bytes(12_500_000, 1_250_000, 1024, 65536, 201)
    assert 0 < lean < full and full - lean >= 4 * 65536 * 1024 * 4


# ---------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_device_generator_equals_numpy_restatement():
    from hassaku_amd.data.synthetic import DeviceInteractions
    from oracle import oracle as orc
    for U, I, skew in ((5000, 3000, 2), (700, 40, 3), (64, 1_000_000, 1)):
        d = DeviceInteractions(U, I, 'cuda', seed=21, skew=skew)
        ptr = d.csr_indptr.cpu().numpy()
        idx = d.csr_indices.cpu().numpy()
        cu = d.coo_user.cpu().numpy()
        deg = orc.synth_degrees(np.arange(U), d.deg_min, d.deg_span, 21)
        assert np.array_equal(np.diff(ptr), deg) and ptr[0] == 0 and ptr[-1] == d.nnz == len(idx)
        assert np.array_equal(cu, np.repeat(np.arange(U), deg))
        users = np.random.RandomState(0).choice(U, size=min(U, 300), replace=False)
        for u, row in zip(users, orc.synth_rows(users, I, d.deg_min, d.deg_span, skew, 21)):
            assert np.array_equal(idx[ptr[u]:ptr[u + 1]], row), (U, I, u)
        assert d.coo_item.data_ptr() == d.csr_indices.data_ptr()


def _softplus(z):
    return np.maximum(z, 0) + np.log1p(np.exp(-np.abs(z)))


class _Tracked:
    """Host copy of the table rows the checked steps touch: (p, m, v) advanced with the oracle's AdamW, every tracked
    row every step (zero gradient when the step does not touch it) -- dense semantics on the tracked subset."""

    def __init__(self, p_dev, m_dev, v_dev, lr, wd):
        self.dev = (p_dev, m_dev, v_dev)
        self.lr, self.wd = lr, wd
        self.rows = np.zeros(0, dtype=np.int64)
        self.P = [np.zeros((0,) + tuple(p_dev.shape[1:]), np.float32) for _ in range(3)]

    def add(self, rows, steps_done, orc):
        new = np.setdiff1d(np.unique(rows), self.rows)
        if len(new) == 0:
            return
        sel = torch.from_numpy(new).cuda()
        fresh = [np.ascontiguousarray(t.index_select(0, sel).cpu().numpy()) for t in self.dev]
        # lazily updated rows: the table still holds their state as of step 0 -- bring them to `steps_done`
        for t in range(1, steps_done + 1):
            orc.adamw_step(fresh[0], None, fresh[1], fresh[2], self.lr, self.wd, t)
        rows_all = np.concatenate([self.rows, new])
        order = np.argsort(rows_all, kind='stable')
        self.rows = rows_all[order]
        self.P = [np.ascontiguousarray(np.concatenate([a, b])[order]) for a, b in zip(self.P, fresh)]

    def index(self, rows):
        i = np.searchsorted(self.rows, rows)
        assert np.array_equal(self.rows[i], rows)
        return i

    def step(self, grad_rows, grads, t, orc):
        g = np.zeros_like(self.P[0])
        g[self.index(grad_rows)] = grads
        orc.adamw_step(self.P[0], g, self.P[1], self.P[2], self.lr, self.wd, t)

    def device_now(self):
        sel = torch.from_numpy(self.rows).cuda()
        return [t.index_select(0, sel).cpu().numpy() for t in self.dev]


def _cfg5_share_worker(rank, world, port, out_dir):
    _init(rank, world, port, 'nccl')
    from conftest import assert_adam_param_close
    from hassaku_amd.data.synthetic import DeviceInteractions
    from hassaku_amd.dist import Comm, ShardedBprMf, init_shard_tables, item_range
    from oracle import oracle as orc
    orc.build()
    dev = torch.device('cuda', 0)
    comm = Comm()
    # one rank's share of eight: 12.5 M of the 100 M users (all of them local to this 1-rank group), the item range
    # rank 3 of 8 owns of the 10 M-item catalogue
    U, I_glob, D, N, G, S = 12_500_000, 10_000_000, 1024, 200, 4096, 3
    lo, hi = item_range(I_glob, 3, 8)
    assert (lo, hi) == (3_750_000, 5_000_000)
    lr, wd = 1e-3, 1e-2
    data = DeviceInteractions(U, I_glob, dev, seed=5)
    assert 19.9 < data.nnz / U < 20.1
    tabs = init_shard_tables(0, 1, U, hi - lo, D, dev, seed=64)
    assert tabs['user_emb'].numel() > 2 ** 31 and tabs['item_emb'].shape == (1_250_000, D)
    std = float(tabs['user_emb'][-4096:].std())
    assert abs(std - 0.1 / D) < 0.02 * 0.1 / D            # the reference's law (train/utils.py:5-13) at the far end too
    sh = ShardedBprMf(comm, tabs['user_emb'], tabs['item_emb'], tabs['item_bias'], None, None, lr=lr, wd=wd, batch=G,
                      n_neg=N, seed=11, inputs_are_shards=True, n_users=U, n_items=I_glob, item_shard=(lo, hi),
                      **data.device_arrays())
    assert sh.sh.base.lazy_items == 1 and sh.sh.base.lazy_users == 1 and sh.sh.base.ws_sharded == 1
    fu, fi = sh.flush_cadence()
    assert 100 <= fu <= 1000 and fi >= 100, (fu, fi)      # size-derived: no sweep of the 169 GB inside these steps
    order = data.random_order(S * G, seed=3)
    tu = _Tracked(sh.user_emb, sh.m['user_emb'], sh.v['user_emb'], lr, wd)
    ti = _Tracked(sh.item_emb, sh.m['item_emb'], sh.v['item_emb'], lr, wd)
    tb = _Tracked(sh.item_bias.view(-1, 1), sh.m['item_bias'].view(-1, 1), sh.v['item_bias'].view(-1, 1), lr, wd)
    # rows no step will touch (checked at the end: they only decay), including the very last rows of both tables
    rs = np.random.RandomState(1)
    idle_u = np.unique(np.concatenate([rs.randint(0, U, size=3000), [U - 1, U - 2, (2 ** 31) // D, (2 ** 31) // D + 1]]))
    idle_i = np.unique(np.concatenate([rs.randint(0, hi - lo, size=3000), [hi - lo - 1]]))
    init_u = sh.user_emb.index_select(0, torch.from_numpy(idle_u).cuda()).cpu().numpy()
    init_i = sh.item_emb.index_select(0, torch.from_numpy(idle_i).cuda()).cpu().numpy()
    inv = np.float32(1.0 / (G * N))
    for s in range(1, S + 1):
        start = (s - 1) * G
        offs, items, ug = (x.cpu().numpy().astype(np.int64) for x in sh.peek_batch(order, start))
        n_ent = int(offs[G])
        items = items[:n_ent]
        assert n_ent > G * N // 10 and items.min() >= 0 and items.max() < hi - lo and ug.max() < U
        pos_item = data.coo_item[order[start:start + G]].cpu().numpy().astype(np.int64)
        assert np.array_equal(ug, data.coo_user[order[start:start + G]].cpu().numpy())
        own_pos = (pos_item >= lo) & (pos_item < hi)
        cnt = np.diff(offs)
        assert np.array_equal(items[offs[:-1][own_pos]], pos_item[own_pos] - lo)   # the positive first, if owned
        for t in (tu, ):
            t.add(ug, s - 1, orc)
        ti.add(items, s - 1, orc)
        tb.add(items, s - 1, orc)
        Ur = tu.P[0][tu.index(ug)]                        # [G, D] current user rows
        eb = np.repeat(np.arange(G), cnt)                 # positive of every kept entry
        ii = ti.index(items)
        Ir = ti.P[0][ii]                                  # [n_ent, D]
        sc = np.einsum('ed,ed->e', Ur[eb], Ir).astype(np.float32) + tb.P[0][ii, 0]
        is_pos = np.zeros(n_ent, dtype=bool)
        is_pos[offs[:-1][own_pos]] = True
        s0 = np.zeros(G, np.float32)
        s0[own_pos] = sc[offs[:-1][own_pos]]
        x = (s0[eb] - sc).astype(np.float32)
        g = np.where(is_pos, np.float32(0), inv / (np.float32(1) + np.exp(x))).astype(np.float32)
        gsum = np.bincount(eb, weights=g.astype(np.float64), minlength=G).astype(np.float32)
        g_entry = np.where(is_pos, -gsum[eb], g).astype(np.float32)
        loss = float((_softplus(-x.astype(np.float64)) * ~is_pos).sum() / (G * N))
        # user-row gradients: per positive, then the positives of one user summed
        dUb = np.zeros((G, D), np.float32)
        nz = cnt > 0
        dUb[nz] = np.add.reduceat(g_entry[:, None] * Ir, offs[:-1][nz], axis=0)
        uu, inv_u = np.unique(ug, return_inverse=True)
        gU = np.zeros((len(uu), D), np.float32)
        np.add.at(gU, inv_u, dUb)
        # item-row gradients: entries grouped by item
        by_item = np.argsort(items, kind='stable')
        it_sorted = items[by_item]
        first = np.flatnonzero(np.r_[True, it_sorted[1:] != it_sorted[:-1]])
        gI = np.add.reduceat(g_entry[by_item, None] * Ur[eb[by_item]], first, axis=0)
        gIb = np.add.reduceat(g_entry[by_item], first)
        tu.step(uu, gU, s, orc)
        ti.step(it_sorted[first], gI, s, orc)
        tb.step(it_sorted[first], gIb[:, None], s, orc)
        # every step names its successor: the next batch is sampled, routed and item-sorted on the side stream under this
        # step (the natively issued step's prefetch, as bench.py and the Trainer drive it).  The tables are not touched by
        # that preparation (HSK_SHARD_AHEAD, the ahead-of-time replay of the next batch's rows, is off by default), so the
        # host model above -- which reads a row's state as of step 0 when it first meets it -- stays exact.
        sh.step_sampled(order, start, next_start=start + G if s < S else None)
        got = sh.last_loss()
        assert abs(got - loss) <= 2e-5 * abs(loss), (s, got, loss)
    sh.flush()
    sh.check_status('cfg5 share')
    for t, name in ((tu, 'user_emb'), (ti, 'item_emb'), (tb, 'item_bias')):
        p, m, v = t.device_now()
        assert_adam_param_close(p, t.P[0], name)
        for got_, ref_, what in ((m, t.P[1], 'exp_avg'), (v, t.P[2], 'exp_avg_sq')):
            scale = np.abs(ref_).max()
            err = np.abs(got_ - ref_)
            assert err.max() <= 1e-5 * scale, (name, what, float(err.max()), float(scale), int((err > 1e-5 * scale).sum()),
                                               int(np.argmax(err.max(axis=-1))), len(t.rows))
    # rows outside every batch: S zero-gradient steps = S multiplications by fl32(1 - lr*wd), moments exactly 0
    decay = np.float32(1.0 - lr * wd)
    for tab, key, rows, init, touched in ((sh.user_emb, 'user_emb', idle_u, init_u, tu.rows),
                                          (sh.item_emb, 'item_emb', idle_i, init_i, ti.rows)):
        keep = ~np.isin(rows, touched)
        sel = torch.from_numpy(rows[keep]).cuda()
        want = init[keep].copy()
        for _ in range(S):
            want = (want * decay).astype(np.float32)
        assert np.array_equal(tab.index_select(0, sel).cpu().numpy(), want), key
        assert float(sh.m[key].index_select(0, sel).abs().max()) == 0.0
        assert float(sh.v[key].index_select(0, sel).abs().max()) == 0.0
    with open(os.path.join(out_dir, 'ok'), 'w') as f:
        f.write('%d %d %d' % (len(tu.rows), len(ti.rows), sh.step_count))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_cfg5_one_rank_share_at_full_size(tmp_path):
    """169 GB of tables + moments on one MI355X: needs (nearly) the whole 288 GB HBM of the card."""
    free, total = torch.cuda.mem_get_info()
    if total < 250 * 2 ** 30:
        pytest.skip('needs a 288 GB device')
    mp.spawn(_cfg5_share_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    n_u, n_i, steps = (int(x) for x in open(tmp_path / 'ok').read().split())
    assert steps == 3 and n_u > 10_000 and n_i > 100_000


def _cfg5_small_worker(rank, world, port, out_dir):
    """the cfg5 construction path (device generator, shard-direct init, no full table anywhere) on two ranks sharing
    the GPU: every rank ends with the same loss, a clean status, and the shards assemble into a model that evaluates"""
    _init(rank, world, port, 'gloo')
    torch.cuda.set_device(0)
    from hassaku_amd.data.synthetic import DeviceInteractions
    from hassaku_amd.dist import Comm, ShardedBprMf, init_shard_tables, local_user_count
    comm = Comm()
    U, I, D, N, B = 30_000, 40_000, 1024, 200, 64
    data = DeviceInteractions(U, I, 'cuda', seed=5)
    tabs = init_shard_tables(rank, world, U, I, D, 'cuda', seed=64)
    assert tabs['user_emb'].shape[0] == local_user_count(U, rank, world)
    sh = ShardedBprMf(comm, tabs['user_emb'], tabs['item_emb'], tabs['item_bias'], None, None, lr=1e-3, wd=1e-4, batch=B,
                      n_neg=N, seed=11, inputs_are_shards=True, n_users=U, n_items=I, flush_every=4,
                      **data.device_arrays())
    order = data.random_order(12 * world * B, seed=3)
    losses = []
    for s in range(12):
        sh.step_sampled(order, s * world * B, next_start=(s + 1) * world * B if s < 11 else None)
        losses.append(sh.last_loss())
    sh.flush()
    sh.check_status()
    full_i, full_ib = sh.gather_item_table()
    assert full_i.shape == (I, D) and torch.isfinite(full_i).all()
    np.save(os.path.join(out_dir, f'loss{rank}.npy'), np.array(losses))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_cfg5_construction_path_two_ranks(tmp_path):
    mp.spawn(_cfg5_small_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    l0, l1 = (np.load(tmp_path / f'loss{r}.npy') for r in (0, 1))
    assert np.array_equal(l0, l1) and np.isfinite(l0).all()
    assert abs(l0[0] - np.log(2.0)) < 0.05 and l0[-1] < l0[0]
