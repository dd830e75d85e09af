"""Multi-process tests of the row-sharded multi-GPU step (hassaku_amd/dist.py).

CPU (world_size 2, gloo): the PROTOCOL -- ownership, routing into fixed-capacity slots, the four collectives,
owner-side accumulation -- executed with the oracle's arithmetic and compared with the un-sharded oracle step.
GPU (2 ranks sharing cuda:0, gloo staging through the host): the real kernels through hsk_mp_*, compared with
the single-GPU fused step on the same global batch, plus the users-sharded evaluation.
"""
import os
import socket
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


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _init(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)


# ---------------------------------------------------------------------------------------------------
# CPU: protocol with oracle arithmetic
# ---------------------------------------------------------------------------------------------------
def _route_host(gu, world, C):
    """host restatement of k_route_requests: stable slot per destination, in batch order"""
    req = -np.ones(world * C, dtype=np.int32)
    slot_of_b = np.zeros(len(gu), dtype=np.int64)
    used = np.zeros(world, dtype=np.int64)
    for b, u in enumerate(gu):
        d = int(u % world)
        assert used[d] < C, 'capacity overflow'
        req[d * C + used[d]] = u // world
        slot_of_b[b] = d * C + used[d]
        used[d] += 1
    return req, slot_of_b


def _cpu_protocol_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    from hassaku_amd.dist import Comm, local_user_count, pair_capacity
    from oracle import oracle as orc
    comm = Comm()
    rng = np.random.RandomState(0)                      # same stream on every rank
    U, I, D, B, N, lr, wd = 37, 50, 16, 12, 5, 1e-2, 1e-3
    Ufull = (rng.randn(U, D) * 0.1).astype(np.float32)
    Iw = (rng.randn(I, D) * 0.1).astype(np.float32)
    Ib = (rng.randn(I) * 0.1).astype(np.float32)
    C = pair_capacity(B, world)
    Uloc = Ufull[rank::world].copy()
    assert Uloc.shape[0] == local_user_count(U, rank, world)
    mU, vU = np.zeros_like(Uloc), np.zeros_like(Uloc)
    mI, vI, mIb, vIb = np.zeros_like(Iw), np.zeros_like(Iw), np.zeros_like(Ib), np.zeros_like(Ib)
    ref = orc.MfOracleTrainer(Ufull, Iw, Ib, lr=lr, wd=wd) if rank == 0 else None
    for step in range(1, 5):
        gu = rng.randint(0, U, size=world * B).astype(np.int64)            # the GLOBAL batch, known to all
        gu[1] = gu[0]                                                       # duplicate users, also across ranks
        gu[B] = gu[0]
        gi = rng.randint(0, I, size=(world * B, N + 1)).astype(np.int64)
        lu, li = gu[rank * B:(rank + 1) * B], gi[rank * B:(rank + 1) * B]
        req, slot_of_b = _route_host(lu, world, C)
        # the owner recomputes the requests it will receive from the global batch every rank knows ...
        req_recv = np.concatenate([_route_host(gu[s * B:(s + 1) * B], world, C)[0][rank * C:(rank + 1) * C]
                                   for s in range(world)])
        # ... which is exactly what exchanging them would deliver
        exchanged = torch.empty(world * C, dtype=torch.int32)
        comm.all_to_all(exchanged, torch.from_numpy(req))
        assert np.array_equal(exchanged.numpy(), req_recv)
        rows_send = np.zeros((world * C, D), np.float32)
        ok = req_recv >= 0
        rows_send[ok] = Uloc[req_recv[ok]]
        rows_recv = torch.empty((world * C, D))
        comm.all_to_all(rows_recv, torch.from_numpy(rows_send))
        Ub = rows_recv.numpy()[slot_of_b]                                   # [B, D] rows of my batch users
        # local compute with the oracle's closed forms; the loss mean is over the GLOBAL batch
        logits = orc.mf_scores(Ub, Iw, Ib, None, None, np.arange(B), li)
        _, g_local = orc.bpr_loss_grad(logits)
        g = g_local / world
        gU, gI, gIb, _, _ = orc.mf_backward(Ub, Iw, np.arange(B), li, g)
        red = torch.from_numpy(np.concatenate([gI.reshape(-1), gIb]))
        comm.all_reduce(red)
        gI, gIb = red.numpy()[:I * D].reshape(I, D), red.numpy()[I * D:]
        grads_send = np.zeros((world * C, D), np.float32)
        grads_send[slot_of_b] = gU                                          # gU is per batch position (identity index)
        grads_recv = torch.empty((world * C, D))
        comm.all_to_all(grads_recv, torch.from_numpy(grads_send))
        # owner: sum the slots of each row in slot order, dense AdamW on the local shard
        gloc = np.zeros_like(Uloc)
        for s in range(world * C):
            if req_recv[s] >= 0:
                gloc[req_recv[s]] += grads_recv.numpy()[s]
        orc.adamw_step(Uloc, gloc, mU, vU, lr, wd, step)
        orc.adamw_step(Iw, np.ascontiguousarray(gI), mI, vI, lr, wd, step)
        orc.adamw_step(Ib, np.ascontiguousarray(gIb), mIb, vIb, lr, wd, step)
        if ref is not None:
            ref.step(gu, gi)
    parts = comm.all_gather(torch.from_numpy(np.pad(Uloc, ((0, local_user_count(U, 0, world) - Uloc.shape[0]), (0, 0)))))
    if rank == 0:
        full = np.empty_like(Ufull)
        for r, p in enumerate(parts):
            full[r::world] = p.numpy()[:local_user_count(U, r, world)]
        np.savez(os.path.join(out_dir, 'res.npz'), U=full, I=Iw, Ib=Ib, rU=ref.P['user_emb'], rI=ref.P['item_emb'],
                 rIb=ref.P['item_bias'])
    dist.destroy_process_group()


def test_sharded_protocol_equals_unsharded_oracle(tmp_path):
    from conftest import assert_adam_param_close
    from oracle import oracle as orc
    orc.build()
    mp.spawn(_cpu_protocol_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = np.load(os.path.join(str(tmp_path), 'res.npz'))
    assert_adam_param_close(r['U'], r['rU'], 'user_emb')
    assert_adam_param_close(r['I'], r['rI'], 'item_emb')
    assert_adam_param_close(r['Ib'], r['rIb'], 'item_bias')


def test_ownership_capacity_and_csr_shards():
    from hassaku_amd.data.csr import UserItemCsr
    from hassaku_amd.dist import local_user_count, owner_of, pair_capacity
    assert [local_user_count(10, r, 4) for r in range(4)] == [3, 3, 2, 2]
    o, l = owner_of(np.arange(10), 4)
    assert list(o) == [0, 1, 2, 3, 0, 1, 2, 3, 0, 1] and list(l) == [0, 0, 0, 0, 1, 1, 1, 1, 2, 2]
    for B, W in ((4096, 8), (4096, 2), (128, 4), (8, 8)):
        C = pair_capacity(B, W)
        assert B / W < C <= B and C % 4 == 0 or C == B
    rng = np.random.RandomState(0)
    pairs = np.argwhere(rng.rand(11, 30) < 0.3)
    csr = UserItemCsr.from_pairs(pairs[:, 0], pairs[:, 1], 11, 30)
    for r in range(3):
        sub = csr.subset_rows(r, 3)
        assert sub.n_rows == local_user_count(11, r, 3)
        for j in range(sub.n_rows):
            assert np.array_equal(sub.row(j), csr.row(r + 3 * j))


# ---------------------------------------------------------------------------------------------------
# GPU: the real kernels, two ranks on one device
# ---------------------------------------------------------------------------------------------------
def _toy_problem():
    rng = np.random.RandomState(3)
    U, I, D, B, N = 211, 300, 64, 48, 12
    pairs = np.argwhere(rng.rand(U, I) < 0.06)
    pairs = pairs[rng.permutation(len(pairs))]
    P = {'user_emb': (rng.randn(U, D) * 0.05).astype(np.float32), 'item_emb': (rng.randn(I, D) * 0.05).astype(np.float32),
         'item_bias': (rng.randn(I) * 0.1).astype(np.float32), 'user_bias': (rng.randn(U) * 0.1).astype(np.float32)}
    val = np.argwhere(rng.rand(U, I) < 0.02)
    return U, I, D, B, N, pairs, P, val


def _gpu_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    torch.cuda.set_device(0)
    from conftest import csr_from_pairs
    from hassaku_amd.data.csr import UserItemCsr
    from hassaku_amd.dist import Comm, ShardedBprMf, evaluate_sharded
    from hassaku_amd.eval.eval import FullEvaluator
    comm = Comm()
    U, I, D, B, N, pairs, P, val = _toy_problem()
    ptr, idx = csr_from_pairs(pairs, U)
    dev = lambda a, dt=None: (torch.from_numpy(np.ascontiguousarray(a)).to(dt) if dt else torch.from_numpy(np.ascontiguousarray(a))).cuda()
    t = {k: dev(v) for k, v in P.items()}
    sh = ShardedBprMf(comm, t['user_emb'], t['item_emb'], t['item_bias'], t['user_bias'], None, lr=2e-3, wd=1e-4,
                      batch=B, n_neg=N, csr_indptr=dev(ptr), csr_indices=dev(idx), coo_user=dev(pairs[:, 0], torch.int32),
                      coo_item=dev(pairs[:, 1], torch.int32), seed=77)
    order = torch.from_numpy(np.random.RandomState(1).permutation(len(pairs))).cuda()
    losses = []
    n_steps = 70                                   # crosses the periodic flush at step 64
    for s in range(n_steps):
        sh.step_sampled(order, (s % 4) * world * B)
        if s % 9 == 0:
            losses.append(sh.last_loss())
    sh.check_status()
    full_u, full_ub = sh.gather_user_table()

    class DS:   # the attributes the evaluators read from a FullEvalDataset
        def device_arrays(self, device):
            lp, li = self.label_csr.to_device(device)
            ep, ei = self.exclude_csr.to_device(device)
            return {'label_indptr': lp, 'label_indices': li, 'excl_indptr': ep, 'excl_indices': ei}
    ds = DS()
    ds.label_csr = UserItemCsr.from_pairs(val[:, 0], val[:, 1], U, I)
    ds.exclude_csr = UserItemCsr.from_pairs(pairs[:, 0], pairs[:, 1], U, I)
    ds._device_cache = {}
    groups = torch.from_numpy((np.arange(U) % 2).astype(np.float32))
    ev = FullEvaluator(aggr_by_group=True, n_groups=2, user_to_user_group=groups)
    metrics = evaluate_sharded(comm, sh, ds, ev, chunk=64)
    from hassaku_amd.dist import evaluate_item_sharded
    m_items = evaluate_item_sharded(comm, full_u, sh.item_emb, sh.item_bias, full_ub, None, ds, ev, chunk=50)
    assert sorted(m_items) == sorted(metrics)
    for k in metrics:   # item-sharded scoring + candidate all-gather + merge == users-sharded scoring
        assert abs(m_items[k] - metrics[k]) < 1e-9, (k, m_items[k], metrics[k])
    if rank == 0:
        np.savez(os.path.join(out_dir, 'mp.npz'), U=full_u.cpu().numpy(), Ub=full_ub.cpu().numpy(),
                 I=sh.item_emb.cpu().numpy(), Ib=sh.item_bias.cpu().numpy(), losses=np.array(losses),
                 metric_names=np.array(sorted(metrics)), metric_values=np.array([metrics[k] for k in sorted(metrics)]))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_sharded_step_equals_single_gpu_step(tmp_path):
    from conftest import assert_adam_param_close, csr_from_pairs
    from hassaku_amd import hip_ops as ops
    from hassaku_amd.data.csr import UserItemCsr
    world = 2
    mp.spawn(_gpu_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = np.load(os.path.join(str(tmp_path), 'mp.npz'))
    # single GPU, global batch world*B, same seed / order / step numbering -> same samples
    U, I, D, B, N, pairs, P, val = _toy_problem()
    ptr, idx = csr_from_pairs(pairs, U)
    dev = lambda a, dt=None: (torch.from_numpy(np.ascontiguousarray(a)).to(dt) if dt else torch.from_numpy(np.ascontiguousarray(a))).cuda()
    t = {k: dev(v) for k, v in P.items()}
    st = ops.BprMfFusedState(t['user_emb'], t['item_emb'], t['item_bias'], t['user_bias'], None, lr=2e-3, wd=1e-4,
                             max_batch=world * B, max_cols=N + 1, seed=77, csr_indptr=dev(ptr), csr_indices=dev(idx),
                             coo_user=dev(pairs[:, 0], torch.int32), coo_item=dev(pairs[:, 1], torch.int32))
    order = torch.from_numpy(np.random.RandomState(1).permutation(len(pairs))).cuda()
    losses = []
    for s in range(70):
        st.step_sampled(order, (s % 4) * world * B, world * B, N)
        if s % 9 == 0:
            losses.append(st.last_loss())
    st.flush()
    st.check_status()
    np.testing.assert_allclose(r['losses'], np.array(losses), rtol=2e-5)
    assert_adam_param_close(r['U'], t['user_emb'].cpu().numpy(), 'user_emb')
    assert_adam_param_close(r['Ub'], t['user_bias'].cpu().numpy(), 'user_bias')
    assert_adam_param_close(r['I'], t['item_emb'].cpu().numpy(), 'item_emb')
    assert_adam_param_close(r['Ib'], t['item_bias'].cpu().numpy(), 'item_bias')
    # users-sharded evaluation == single-GPU evaluation of the gathered tables
    lab = UserItemCsr.from_pairs(val[:, 0], val[:, 1], U, I)
    exc = UserItemCsr.from_pairs(pairs[:, 0], pairs[:, 1], U, I)
    lp, li = lab.to_device('cuda')
    ep, ei = exc.to_device('cuda')
    ks = [100, 50, 10, 5]
    u = torch.arange(U, device='cuda')
    _, ids, _ = ops.mf_eval_topk(dev(r['U']), dev(r['I']), dev(r['Ib']), dev(r['Ub']), None, u, 100, ep, ei)
    met = ops.rank_metrics(ids, u, lp, li, ks).double().cpu().numpy()
    got = dict(zip([str(x) for x in r['metric_names']], r['metric_values']))
    grp = np.arange(U) % 2
    for tt, k in enumerate(ks):
        for j, name in enumerate(('precision', 'recall', 'ndcg')):
            assert abs(got[f'{name}@{k}'] - met[:, tt, j].mean()) < 1e-9, (name, k)
            assert abs(got[f'group_1_{name}@{k}'] - met[grp == 1, tt, j].mean()) < 1e-9, (name, k)


def _trainer_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0')
    from hassaku_amd.algorithms.algorithms_utils import AlgorithmsEnum
    from hassaku_amd.data.data_utils import DatasetsEnum
    from hassaku_amd.experiment_helper import run_train_val
    conf = {'data_path': os.path.join(out_dir, 'data'), 'model_save_path': os.path.join(out_dir, 'models'),
            'embedding_dim': 32, 'lr': 5e-3, 'wd': 1e-5, 'use_user_bias': False, 'use_item_bias': True,
            'use_global_bias': False, 'optimizer': 'adamw', 'n_epochs': 3, 'max_patience': 2, 'train_batch_size': 64,
            'neg_train': 8, 'rec_loss': 'bpr', 'eval_batch_size': 64, 'device': 'cuda',
            'running_settings': {'use_wandb': False, 'batch_verbose': False, 'dist_backend': 'gloo'}}
    best, conf = run_train_val(AlgorithmsEnum.mf, DatasetsEnum.ml100k, conf)
    import json
    json.dump({'best': best, 'model_path': conf['model_path']}, open(os.path.join(out_dir, f'rank{rank}.json'), 'w'))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_run_train_val(tmp_path):
    """`torchrun --nproc-per-node 2 run_experiment.py` equivalent: both ranks agree on the metrics, rank 0 wrote a
    complete model.pth (user table gathered from the shards) that evaluates to the same validation metrics."""
    import json
    from hassaku_amd.data.synthetic import generate, write_csv_dataset
    ds_path = str(tmp_path / 'data' / 'ml100k' / 'processed_dataset')
    write_csv_dataset(generate(200, 300, 8000, seed=4, n_groups=2), ds_path)
    mp.spawn(_trainer_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (json.load(open(tmp_path / f'rank{r}.json')) for r in (0, 1))
    assert r0['best'] == r1['best'] and r0['model_path'] == r1['model_path']
    assert r0['best']['best_epoch'] >= 0 and r0['best']['ndcg@10'] > 0.01
    assert os.path.isfile(os.path.join(r0['model_path'], 'model.pth')) and os.path.isfile(os.path.join(r0['model_path'], 'conf.yml'))
    from hassaku_amd.algorithms.sgd_alg import SGDMatrixFactorization
    from hassaku_amd.data.data_utils import get_dataloader
    from hassaku_amd.eval.eval import FullEvaluator, evaluate_recommender_algorithm
    loader = get_dataloader({'dataset_path': ds_path, 'eval_batch_size': 64, 'running_settings': {}}, 'val')
    model = SGDMatrixFactorization(200, 300, 32, False, True, False).to('cuda')
    model.load_model_from_path(r0['model_path'])
    ev = FullEvaluator(True, loader.dataset.n_user_groups, loader.dataset.user_to_user_group)
    m = evaluate_recommender_algorithm(model, loader, ev, 'cuda')
    for k in ('ndcg@10', 'recall@50', 'group_0_precision@5'):
        assert abs(m[k] - r0['best'][k]) < 1e-6, k   # fp32 per-batch sums here vs fp64 sums in the sharded path
