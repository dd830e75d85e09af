"""Multi-process tests of the item-sharded multi-GPU step and evaluation (hassaku_amd/dist.py, csrc/hsk_shard.inc).

CPU (world_size 2, gloo): the PROTOCOL -- item-range / user-row ownership, the kept-entries rule, slot routing, the four
collectives (all_gather of user rows, all_reduce of the positive scores, all_reduce of the negatives' weights,
reduce_scatter of the user-row gradients), owner-side accumulation -- executed with the oracle's arithmetic and
compared with the un-sharded oracle step.
GPU: the real kernels through hsk_shard_*: (a) ONE rank on backend nccl (= RCCL): every collective the step and the
evaluation issue runs on RCCL with async_op / wait ordering against the kernels; (b) two ranks sharing cuda:0 (gloo
staging through the host): the multi-rank semantics.  Both are compared with the single-GPU fused step on the same
global batch, and the item-sharded evaluation with the single-GPU evaluation of the gathered tables.
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


def _init(rank, world, port, backend='gloo'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend == 'nccl':
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)


# ---------------------------------------------------------------------------------------------------
# CPU: protocol with oracle arithmetic
# ---------------------------------------------------------------------------------------------------
def _route_host(gu, world, C):
    """host restatement of k_shard_route: slot = owner*C + position among that owner's positives, in batch order"""
    slot_of_b = np.zeros(len(gu), dtype=np.int64)
    req = -np.ones((world, C), dtype=np.int64)
    used = np.zeros(world, dtype=np.int64)
    for b, u in enumerate(gu):
        d = int(u % world)
        assert used[d] < C, 'capacity overflow'
        req[d, used[d]] = u // world
        slot_of_b[b] = d * C + used[d]
        used[d] += 1
    return req, slot_of_b


def _softplus(z):
    return np.maximum(z, 0) + np.log1p(np.exp(-np.abs(z)))


def _cpu_protocol_worker(rank, world, port, out_dir, loss='bpr'):
    _init(rank, world, port)
    from hassaku_amd.dist import Comm, item_range, local_user_count, user_capacity
    from oracle import oracle as orc
    comm = Comm()
    rng = np.random.RandomState(0)                      # same stream on every rank
    U, I, D, B, N, lr, wd = 37, 50, 16, 12, 5, 1e-2, 1e-3
    G = world * B
    log_adjust = float(np.log(I / N)) if loss == 'sampled_softmax' else 0.0
    Ufull = (rng.randn(U, D) * 0.1).astype(np.float32)
    Ifull = (rng.randn(I, D) * 0.1).astype(np.float32)
    Ibfull = (rng.randn(I) * 0.1).astype(np.float32)
    C = user_capacity(0.6, G)
    lo, hi = item_range(I, rank, world)
    Uloc, Iloc, Ibloc = Ufull[rank::world].copy(), Ifull[lo:hi].copy(), Ibfull[lo:hi].copy()
    assert Uloc.shape[0] == local_user_count(U, rank, world)
    mU, vU = np.zeros_like(Uloc), np.zeros_like(Uloc)
    mI, vI, mIb, vIb = np.zeros_like(Iloc), np.zeros_like(Iloc), np.zeros_like(Ibloc), np.zeros_like(Ibloc)
    ref = orc.MfOracleTrainer(Ufull, Ifull, Ibfull, lr=lr, wd=wd, loss=loss, log_adjust=log_adjust) if rank == 0 else None
    losses = []
    n_coll = {'s0': 0, 'gsum': 0, 'ssm': 0}
    for step in range(1, 5):
        gu = rng.randint(0, U, size=G).astype(np.int64)                    # the GLOBAL batch, known to all
        gu[1] = gu[0]                                                       # duplicate users, also across slices
        gu[B] = gu[0]
        gi = rng.randint(0, I, size=(G, N + 1)).astype(np.int64)
        req, slot_of_b = _route_host(gu, world, C)
        # owner -> everyone: the batch's user rows
        send = np.zeros((C, D), np.float32)
        ok = req[rank] >= 0
        send[ok] = Uloc[req[rank][ok]]
        rows_all = torch.empty((world * C, D))
        comm.all_gather_into(rows_all, torch.from_numpy(send))
        Ub = rows_all.numpy()[slot_of_b]                                    # [G, D]
        mine = (gi >= lo) & (gi < hi)                                       # the entries this rank keeps
        # positive scores by the positive item's owner
        s0 = np.zeros(G, np.float32)
        own_pos = mine[:, 0]
        s0[own_pos] = (Ub[own_pos] * Iloc[gi[own_pos, 0] - lo]).sum(-1) + Ibloc[gi[own_pos, 0] - lo]
        if loss == 'bpr':                                                   # bce: the owner's own s0 is all it needs
            s0_t = torch.from_numpy(s0)
            comm.all_reduce(s0_t)
            s0 = s0_t.numpy()
            n_coll['s0'] += 1
        inv = np.float32({'bpr': 1.0 / (G * N), 'bce': 1.0 / (G * (N + 1)), 'sampled_softmax': 1.0 / G}[loss])
        dU = np.zeros((world * C, D), np.float32)
        gsum = np.zeros(G, np.float32)
        gI, gIb = np.zeros_like(Iloc), np.zeros_like(Ibloc)
        loss_part = 0.0
        g_entry = np.zeros((G, N + 1), np.float32)
        m_r = np.full(G, -np.inf, np.float32)
        l_r = np.zeros(G, np.float32)
        z_entry = np.zeros((G, N + 1), np.float32)
        # owned negatives: weights, partial user-row gradients, partial sums, partial loss
        for b in range(G):
            ks = [k for k in range(1, N + 1) if mine[b, k]]
            sc = {k: np.float32(np.dot(Ub[b], Iloc[gi[b, k] - lo]) + Ibloc[gi[b, k] - lo]) for k in ks}
            if loss == 'sampled_softmax' and ks:
                z = {k: np.float32(sc[k] + log_adjust) for k in ks}
                m_r[b] = max(z.values())
                for k in ks:
                    e = np.exp(z[k] - m_r[b])
                    l_r[b] += e
                    dU[slot_of_b[b]] += e * Iloc[gi[b, k] - lo]            # unnormalised, scaled below
                    z_entry[b, k] = z[k]
            for k in ks:
                j = gi[b, k] - lo
                if loss == 'bpr':
                    x = s0[b] - sc[k]
                    g = inv / (1.0 + np.exp(x))
                    gsum[b] += g
                    loss_part += _softplus(-x)
                elif loss == 'bce':
                    g = inv / (1.0 + np.exp(-sc[k]))
                    loss_part += _softplus(sc[k])
                else:
                    continue
                g_entry[b, k] = g
                dU[slot_of_b[b]] += g * Iloc[j]
        if loss == 'bpr':
            gs_t = torch.from_numpy(gsum)
            comm.all_reduce(gs_t)
            gsum = gs_t.numpy()
            n_coll['gsum'] += 1
            for b in np.nonzero(own_pos)[0]:
                g_entry[b, 0] = -gsum[b]
                dU[slot_of_b[b]] += -gsum[b] * Iloc[gi[b, 0] - lo]
        elif loss == 'bce':
            for b in np.nonzero(own_pos)[0]:
                g0 = -inv / (1.0 + np.exp(s0[b]))
                g_entry[b, 0] = g0
                dU[slot_of_b[b]] += g0 * Iloc[gi[b, 0] - lo]
                loss_part += _softplus(-s0[b])
        else:
            # one all_gather of the (max, normaliser, s0) triples in place of the two reductions
            trip = torch.from_numpy(np.concatenate([m_r, l_r, s0]))
            all_t = torch.empty(world * 3 * G)
            comm.all_gather_into(all_t, trip)
            n_coll['ssm'] += 1
            t = all_t.numpy().reshape(world, 3, G)
            s0g = t[:, 2, :].sum(0)
            M = np.maximum(t[:, 0, :].max(0), s0g)
            with np.errstate(invalid='ignore'):
                L = np.exp(s0g - M) + (t[:, 1, :] * np.exp(t[:, 0, :] - M)).sum(0)
            for b in range(G):
                scale = (np.exp(m_r[b] - M[b]) if np.isfinite(m_r[b]) else 0.0) * inv / L[b]
                dU[slot_of_b[b]] *= np.float32(scale)
                for k in range(1, N + 1):
                    if mine[b, k]:
                        g_entry[b, k] = np.exp(z_entry[b, k] - M[b]) * inv / L[b]
                if own_pos[b]:
                    g0 = np.exp(s0g[b] - M[b]) * inv / L[b] - inv
                    g_entry[b, 0] = g0
                    dU[slot_of_b[b]] += g0 * Iloc[gi[b, 0] - lo]
                    loss_part += -s0g[b] + M[b] + np.log(L[b])
        grads_mine = torch.empty((C, D))
        comm.reduce_scatter(grads_mine, torch.from_numpy(dU))
        # local item gradient (never leaves the rank) + AdamW on the shard
        for b in range(G):
            for k in range(N + 1):
                if mine[b, k]:
                    gI[gi[b, k] - lo] += g_entry[b, k] * Ub[b]
                    gIb[gi[b, k] - lo] += g_entry[b, k]
        orc.adamw_step(Iloc, gI, mI, vI, lr, wd, step)
        orc.adamw_step(Ibloc, gIb, mIb, vIb, lr, wd, step)
        # owner: sum the slots of each row in slot order, dense AdamW on the local user shard
        gloc = np.zeros_like(Uloc)
        for s in range(C):
            if req[rank][s] >= 0:
                gloc[req[rank][s]] += grads_mine.numpy()[s]
        orc.adamw_step(Uloc, gloc, mU, vU, lr, wd, step)
        lt = torch.tensor([loss_part * float(inv)], dtype=torch.float64)
        comm.all_reduce(lt)
        losses.append(float(lt.item()))
        if ref is not None:
            assert abs(ref.step(gu, gi)[0] - losses[-1]) < 2e-6 * abs(losses[-1]), (loss, ref.step, losses)
    # what each loss costs in scalar collectives per step (hassaku_hip.h, "Losses")
    assert n_coll == {'bpr': {'s0': 4, 'gsum': 4, 'ssm': 0}, 'bce': {'s0': 0, 'gsum': 0, 'ssm': 0},
                      'sampled_softmax': {'s0': 0, 'gsum': 0, 'ssm': 4}}[loss]
    pad = lambda a, n: np.pad(a, ((0, n - a.shape[0]),) + ((0, 0),) * (a.ndim - 1))
    parts = comm.all_gather(torch.from_numpy(pad(Uloc, local_user_count(U, 0, world))))
    n_max = max(item_range(I, r, world)[1] - item_range(I, r, world)[0] for r in range(world))
    iparts = comm.all_gather(torch.from_numpy(pad(Iloc, n_max)))
    bparts = comm.all_gather(torch.from_numpy(pad(Ibloc, n_max)))
    if rank == 0:
        full = np.empty_like(Ufull)
        for r, p in enumerate(parts):
            full[r::world] = p.numpy()[:local_user_count(U, r, world)]
        sizes = [item_range(I, r, world)[1] - item_range(I, r, world)[0] for r in range(world)]
        fi = np.concatenate([p.numpy()[:n] for p, n in zip(iparts, sizes)])
        fb = np.concatenate([p.numpy()[:n] for p, n in zip(bparts, sizes)])
        np.savez(os.path.join(out_dir, 'res.npz'), U=full, I=fi, Ib=fb, rU=ref.P['user_emb'], rI=ref.P['item_emb'],
                 rIb=ref.P['item_bias'])
    dist.destroy_process_group()


@pytest.mark.parametrize('loss', ['bpr', 'bce', 'sampled_softmax'])
def test_sharded_protocol_equals_unsharded_oracle(tmp_path, loss):
    """the three losses the reference's DataParallel wrap carries (train/rec_losses.py:27-139): bpr with its two scalar
    reductions, bce with none, sampled softmax with one all_gather of (max, normaliser, s0) triples"""
    from conftest import assert_adam_param_close
    from oracle import oracle as orc
    orc.build()
    mp.spawn(_cpu_protocol_worker, args=(2, _free_port(), str(tmp_path), loss), nprocs=2, join=True)
    r = np.load(os.path.join(str(tmp_path), 'res.npz'))
    assert_adam_param_close(r['U'], r['rU'], 'user_emb')
    assert_adam_param_close(r['I'], r['rI'], 'item_emb')
    assert_adam_param_close(r['Ib'], r['rIb'], 'item_bias')


def test_ownership_capacities_and_csr_shards():
    from hassaku_amd.data.csr import UserItemCsr
    from hassaku_amd.dist import entry_capacity, item_range, local_user_count, owner_of, user_capacity
    assert [local_user_count(10, r, 4) for r in range(4)] == [3, 3, 2, 2]
    o, l = owner_of(np.arange(10), 4)
    assert list(o) == [0, 1, 2, 3, 0, 1, 2, 3, 0, 1] and list(l) == [0, 0, 0, 0, 1, 1, 1, 1, 2, 2]
    for I, W in ((10677, 8), (131072, 8), (7, 3), (10_000_000, 8)):
        ranges = [item_range(I, r, W) for r in range(W)]
        assert ranges[0][0] == 0 and ranges[-1][1] == I
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        assert max(h - l for l, h in ranges) - min(h - l for l, h in ranges) <= 1
    for G, W in ((32768, 8), (8192, 2), (512, 4), (8, 8), (128, 1)):
        C = user_capacity(1.0 / W, G)
        assert G / W <= C <= G and (C % 4 == 0 or C == G)
        cap = entry_capacity(1.0 / W, G, 100)
        assert G * 100 / W < cap <= G * 101
    assert user_capacity(1.0, 128) == 128 and entry_capacity(1.0, 128, 10) == 128 * 11
    # the cfg5 shard fits the item sort on its own: 10 M items over 8 ranks = 1.25 M per shard (cap 4.19 M per device)
    from hassaku_amd import _lib
    if os.path.isfile(_lib.LIB_PATH):
        lib = _lib.load()
        assert lib.hsk_bprmf_workspace_bytes(1000, 10_000_000 // 8, 1024, 8 * 1024, 201) > 0
        assert lib.hsk_bprmf_workspace_bytes(1000, 10_000_000, 1024, 8 * 1024, 201) < 0
    rng = np.random.RandomState(0)
    pairs = np.argwhere(rng.rand(11, 30) < 0.3)
    csr = UserItemCsr.from_pairs(pairs[:, 0], pairs[:, 1], 11, 30)
    for r in range(3):
        sub = csr.subset_rows(r, 3)
        assert sub.n_rows == local_user_count(11, r, 3)
        for j in range(sub.n_rows):
            assert np.array_equal(sub.row(j), csr.row(r + 3 * j))


# ---------------------------------------------------------------------------------------------------
# GPU: the real kernels
# ---------------------------------------------------------------------------------------------------
def _toy_problem(D=64, N=12, U=211, I=300, B=48):
    rng = np.random.RandomState(3)
    pairs = np.argwhere(rng.rand(U, I) < 0.06)
    pairs = pairs[rng.permutation(len(pairs))]
    P = {'user_emb': (rng.randn(U, D) * 0.05).astype(np.float32), 'item_emb': (rng.randn(I, D) * 0.05).astype(np.float32),
         'item_bias': (rng.randn(I) * 0.1).astype(np.float32), 'user_bias': (rng.randn(U) * 0.1).astype(np.float32)}
    val = np.argwhere(rng.rand(U, I) < 0.02)
    return U, I, D, B, N, pairs, P, val


def _dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t.to(dt) if dt else t).cuda()


class _EvalDs:   # the attributes the evaluators read from a FullEvalDataset
    def __init__(self, pairs, val, U, I):
        from hassaku_amd.data.csr import UserItemCsr
        self.label_csr = UserItemCsr.from_pairs(val[:, 0], val[:, 1], U, I)
        self.exclude_csr = UserItemCsr.from_pairs(pairs[:, 0], pairs[:, 1], U, I)
        self.n_users, self.n_items = U, I
        self._device_cache = {}

    def device_arrays(self, device):
        lp, li = self.label_csr.to_device(device)
        ep, ei = self.exclude_csr.to_device(device)
        return {'label_indptr': lp, 'label_indices': li, 'excl_indptr': ep, 'excl_indices': ei}


def _log_adjust(loss, I, N):
    return float(np.log(I / N)) if loss == 'sampled_softmax' else 0.0


def _shard_worker(rank, world, port, out_dir, backend, shape, n_steps, lazy_items, prefetch, native='auto', loss='bpr'):
    _init(rank, world, port, backend)
    torch.cuda.set_device(0)
    from conftest import csr_from_pairs
    from hassaku_amd.dist import Comm, ShardedBprMf, evaluate_item_sharded
    from hassaku_amd.eval.eval import FullEvaluator
    comm = Comm()
    assert comm.native == (backend == 'nccl')
    U, I, D, B, N, pairs, P, val = _toy_problem(**shape)
    ptr, idx = csr_from_pairs(pairs, U)
    t = {k: _dev(v) for k, v in P.items()}
    if loss == 'bce':
        t['user_bias'] = None      # bce sends gradient to the user bias: not carried by the fused / sharded steps
    sh = ShardedBprMf(comm, t['user_emb'], t['item_emb'], t['item_bias'], t['user_bias'], None, lr=2e-3, wd=1e-4,
                      batch=B, n_neg=N, csr_indptr=_dev(ptr), csr_indices=_dev(idx), coo_user=_dev(pairs[:, 0], torch.int32),
                      coo_item=_dev(pairs[:, 1], torch.int32), seed=77, lazy_items=lazy_items, prefetch=prefetch,
                      native=native, loss=loss, log_adjust=_log_adjust(loss, I, N))
    if native == 'host-staged':
        assert sh.issued_natively and sh.backend() == 'host-staged'   # hsk_shard_step itself, not the phased sequence
    elif native is False:
        assert not sh.issued_natively
    order = torch.from_numpy(np.random.RandomState(1).permutation(len(pairs))).cuda()
    G = world * B
    n_slices = len(pairs) // G
    losses = []
    for s in range(n_steps):
        nxt = ((s + 1) % n_slices) * G if s + 1 < n_steps else None
        if s == 5:
            nxt = 0                                 # a wrong guess: the prepared batch must be discarded
        sh.step_sampled(order, (s % n_slices) * G, next_start=nxt)
        if s % 9 == 0:
            losses.append(sh.last_loss())
    sh.check_status()
    # kept entries of the last batch: every rank keeps exactly the global sample's entries inside its item range
    offs, items, ug = (x.cpu().numpy() for x in sh.last_batch())
    full_u, full_ub = sh.gather_user_table()
    full_i, full_ib = sh.gather_item_table()
    ds = _EvalDs(pairs, val, U, I)
    groups = torch.from_numpy((np.arange(U) % 2).astype(np.float32))
    ev = FullEvaluator(aggr_by_group=True, n_groups=2, user_to_user_group=groups)
    metrics = evaluate_item_sharded(comm, sh, ds, ev, chunk=64)
    np.savez(os.path.join(out_dir, f'kept{rank}.npz'), offs=offs, items=items, ug=ug, lo=sh.item_lo, hi=sh.item_hi)
    if rank == 0:
        np.savez(os.path.join(out_dir, 'mp.npz'), U=full_u.cpu().numpy(),
                 Ub=np.zeros(U, np.float32) if full_ub is None else full_ub.cpu().numpy(),
                 I=full_i.cpu().numpy(), Ib=full_ib.cpu().numpy(), losses=np.array(losses),
                 metric_names=np.array(sorted(metrics)), metric_values=np.array([metrics[k] for k in sorted(metrics)]))
    sh.close()
    dist.destroy_process_group()


def _single_gpu_reference(world, shape, n_steps, lazy_items='auto', loss='bpr'):
    """The same global batches through the single-GPU fused step: same seed / order / step numbering -> same samples."""
    from conftest import csr_from_pairs
    from hassaku_amd import hip_ops as ops
    U, I, D, B, N, pairs, P, val = _toy_problem(**shape)
    ptr, idx = csr_from_pairs(pairs, U)
    t = {k: _dev(v) for k, v in P.items()}
    if loss == 'bce':
        t['user_bias'] = None
    G = world * B
    st = ops.BprMfFusedState(t['user_emb'], t['item_emb'], t['item_bias'], t['user_bias'], None, lr=2e-3, wd=1e-4,
                             max_batch=G, max_cols=N + 1, seed=77, csr_indptr=_dev(ptr), csr_indices=_dev(idx),
                             coo_user=_dev(pairs[:, 0], torch.int32), coo_item=_dev(pairs[:, 1], torch.int32),
                             lazy_items=lazy_items, loss=loss, log_adjust=_log_adjust(loss, I, N))
    order = torch.from_numpy(np.random.RandomState(1).permutation(len(pairs))).cuda()
    n_slices = len(pairs) // G
    losses, batches = [], []
    for s in range(n_steps):
        st.step_sampled(order, (s % n_slices) * G, G, N)
        if s % 9 == 0:
            losses.append(st.last_loss())
        bu, bi = st.last_batch(G, N + 1)      # the global batch the device sampler drew for this step
        batches.append((bu.cpu().numpy(), bi.cpu().numpy()))
    st.flush()
    st.check_status()
    lu, li = st.last_batch(G, N + 1)
    return t, np.array(losses), lu.cpu().numpy(), li.cpu().numpy(), batches


def _oracle_replay(shape, batches, loss='bpr'):
    """The same global batches through the CPU oracle (dense AdamW on every row, every step)."""
    from oracle import oracle as orc
    U, I, D, B, N, pairs, P, val = _toy_problem(**shape)
    tr = orc.MfOracleTrainer(P['user_emb'], P['item_emb'], P['item_bias'], None if loss == 'bce' else P['user_bias'], None,
                             lr=2e-3, wd=1e-4, loss=loss, log_adjust=_log_adjust(loss, I, N))
    losses = []
    for s, (u, it) in enumerate(batches):
        loss, _, _, _ = tr.step(u.astype(np.int64), it.astype(np.int64))
        if s % 9 == 0:
            losses.append(float(loss))
    return tr.P, np.array(losses)


def _check_against_single_gpu(tmp_path, world, shape, n_steps, loss='bpr'):
    from conftest import assert_adam_param_close
    from hassaku_amd import hip_ops as ops
    from hassaku_amd.data.csr import UserItemCsr
    r = np.load(os.path.join(str(tmp_path), 'mp.npz'))
    t, losses, lu, li, batches = _single_gpu_reference(world, shape, n_steps, loss=loss)
    has_ub = loss != 'bce'
    U, I, D, B, N, pairs, P, val = _toy_problem(**shape)
    # the samples: each rank kept exactly the entries of the single-GPU batch that fall into its item range,
    # positive first, negatives in column order
    for rank in range(world):
        k = np.load(os.path.join(str(tmp_path), f'kept{rank}.npz'))
        lo, hi = int(k['lo']), int(k['hi'])
        assert np.array_equal(k['ug'], lu)
        for b in range(world * B):
            want = [x - lo for x in li[b] if lo <= x < hi]
            assert list(k['items'][k['offs'][b]:k['offs'][b + 1]]) == want, (rank, b)
    np.testing.assert_allclose(r['losses'], losses, rtol=2e-5)
    assert_adam_param_close(r['U'], t['user_emb'].cpu().numpy(), 'user_emb')
    if has_ub:
        assert_adam_param_close(r['Ub'], t['user_bias'].cpu().numpy(), 'user_bias')
    assert_adam_param_close(r['I'], t['item_emb'].cpu().numpy(), 'item_emb')
    assert_adam_param_close(r['Ib'], t['item_bias'].cpu().numpy(), 'item_bias')
    # ... and DIRECTLY against the oracle: the batches above (each rank's kept entries were just shown to be exactly
    # their entries in its item range) replayed through the CPU restatement of the reference's dense step
    oP, olosses = _oracle_replay(shape, batches, loss)
    np.testing.assert_allclose(r['losses'], olosses, rtol=2e-5)
    assert_adam_param_close(r['U'], oP['user_emb'], 'user_emb vs oracle')
    assert_adam_param_close(r['I'], oP['item_emb'], 'item_emb vs oracle')
    assert_adam_param_close(r['Ib'], oP['item_bias'], 'item_bias vs oracle')
    if has_ub:
        assert_adam_param_close(r['Ub'], oP['user_bias'], 'user_bias vs oracle')
    # item-sharded evaluation == single-GPU evaluation of the gathered tables
    lab = UserItemCsr.from_pairs(val[:, 0], val[:, 1], U, I)
    exc = UserItemCsr.from_pairs(pairs[:, 0], pairs[:, 1], U, I)
    lp, lix = lab.to_device('cuda')
    ep, ei = exc.to_device('cuda')
    ks = [100, 50, 10, 5]
    u = torch.arange(U, device='cuda')
    _, ids, _ = ops.mf_eval_topk(_dev(r['U']), _dev(r['I']), _dev(r['Ib']), _dev(r['Ub']) if has_ub else None, None, u, 100,
                                 ep, ei)
    met = ops.rank_metrics(ids, u, lp, lix, ks).double().cpu().numpy()
    got = dict(zip([str(x) for x in r['metric_names']], r['metric_values']))
    grp = np.arange(U) % 2
    for tt, k in enumerate(ks):
        for j, name in enumerate(('precision', 'recall', 'ndcg')):
            assert abs(got[f'{name}@{k}'] - met[:, tt, j].mean()) < 1e-9, (name, k)
            assert abs(got[f'group_1_{name}@{k}'] - met[grp == 1, tt, j].mean()) < 1e-9, (name, k)


@pytest.mark.gpu
def test_one_rank_rccl_sharded_step_equals_single_gpu_step(tmp_path):
    """backend nccl with ONE rank: all_gather_into / all_reduce / reduce_scatter / all_to_all of the step and the
    evaluation run on RCCL itself (async_op + wait against kernels on the current stream, the side-stream prefetch),
    70 steps across the periodic flush; results = the single-GPU step."""
    shape = dict(D=64, N=12)
    mp.spawn(_shard_worker, args=(1, _free_port(), str(tmp_path), 'nccl', shape, 70, 'auto', True), nprocs=1, join=True)
    _check_against_single_gpu(tmp_path, 1, shape, 70)


def _native_or_phased_worker(rank, world, port, out_dir, native):
    os.makedirs(out_dir, exist_ok=True)
    _shard_worker(rank, world, port, out_dir, 'nccl', dict(D=64, N=12), 70, 'auto', True, native)


@pytest.mark.gpu
def test_native_rccl_step_equals_phased_step_bitwise(tmp_path):
    """hsk_shard_step (the whole step from one C call, ncclAllGather / ncclAllReduce / ncclReduceScatter on the
    library's own communicator, exchanges on a communication stream) == the phase-by-phase sequence driven from Python
    with torch.distributed: same kernels, same order -> the same bits (tables, losses, metrics), 70 steps with the
    side-stream preparation and a wrong next-batch guess."""
    from hassaku_amd import _lib
    if not _lib.load().hsk_rccl_available():
        pytest.skip('librccl not loadable')
    res = []
    for native in (True, False):
        d = str(tmp_path / ('native' if native else 'phased'))
        mp.spawn(_native_or_phased_worker, args=(1, _free_port(), d, native), nprocs=1, join=True)
        res.append(np.load(os.path.join(d, 'mp.npz')))
    for k in ('U', 'Ub', 'I', 'Ib', 'losses', 'metric_values'):
        assert np.array_equal(res[0][k], res[1][k]), k
    _check_against_single_gpu(tmp_path / 'native', 1, dict(D=64, N=12), 70)


@pytest.mark.gpu
def test_two_rank_sharded_step_equals_single_gpu_step(tmp_path):
    shape = dict(D=64, N=12)
    mp.spawn(_shard_worker, args=(2, _free_port(), str(tmp_path), 'gloo', shape, 70, 'auto', True), nprocs=2, join=True)
    _check_against_single_gpu(tmp_path, 2, shape, 70)


@pytest.mark.gpu
def test_three_rank_lazy_items_no_prefetch(tmp_path):
    """odd world size (uneven item ranges, uneven user shards), lazy item AdamW on the shards, D = 402 (float2 rows)"""
    shape = dict(D=402, N=7, U=97, I=1000, B=20)
    mp.spawn(_shard_worker, args=(3, _free_port(), str(tmp_path), 'gloo', shape, 66, True, False), nprocs=3, join=True)
    _check_against_single_gpu(tmp_path, 3, shape, 66)


@pytest.mark.gpu
@pytest.mark.parametrize('world,shape,n_steps,lazy_items,prefetch', [
    (2, dict(D=64, N=12), 70, 'auto', True),
    (3, dict(D=402, N=7, U=97, I=1000, B=20), 66, True, True),
])
def test_native_step_runs_multi_rank_over_host_staged_collectives(tmp_path, world, shape, n_steps, lazy_items, prefetch):
    """hsk_shard_step ITSELF at world 2 and 3 on one GPU: the ranks are processes sharing cuda:0, the collective table is
    the library's host-staged one (RCCL refuses two ranks of a communicator on one device).  What only shows at W >= 2 --
    per-rank counts at C*D, the fork order, one collective table used from two streams, the batch prepared a step ahead
    on the side stream, a wrong next-batch guess -- runs here and is held to the single-GPU step and to the oracle."""
    mp.spawn(_shard_worker, args=(world, _free_port(), str(tmp_path), 'gloo', shape, n_steps, lazy_items, prefetch,
                                  'host-staged'), nprocs=world, join=True)
    _check_against_single_gpu(tmp_path, world, shape, n_steps)


@pytest.mark.gpu
@pytest.mark.parametrize('loss', ['bce', 'sampled_softmax'])
@pytest.mark.parametrize('native', [False, 'host-staged'])
def test_two_rank_sharded_step_other_losses(tmp_path, loss, native):
    """bce (no scalar collective) and sampled softmax (one all_gather of (max, normaliser, s0) triples) through the sharded
    step, phase by phase over gloo and from hsk_shard_step over the host-staged table: == the single-GPU fused step with the
    same loss on the same global batches, and == the oracle's replay (train/rec_losses.py:27-53, :91-139)."""
    shape = dict(D=64, N=12)
    mp.spawn(_shard_worker, args=(2, _free_port(), str(tmp_path), 'gloo', shape, 40, 'auto', True, native, loss),
             nprocs=2, join=True)
    _check_against_single_gpu(tmp_path, 2, shape, 40, loss)


def _overflow_worker(rank, world, port, out_dir, native):
    _init(rank, world, port, 'gloo')
    torch.cuda.set_device(0)
    from conftest import csr_from_pairs
    from hassaku_amd.dist import Comm, ShardedBprMf
    comm = Comm()
    U, I, D, B, N, pairs, P, val = _toy_problem(D=64, N=12)
    ptr, idx = csr_from_pairs(pairs, U)
    G = world * B
    order = torch.from_numpy(np.random.RandomState(1).permutation(len(pairs))).cuda()
    n_slices = len(pairs) // G

    def build(entry_cap):
        t = {k: _dev(v) for k, v in P.items()}
        return ShardedBprMf(comm, t['user_emb'], t['item_emb'], t['item_bias'], t['user_bias'], None, lr=2e-3, wd=1e-4,
                            batch=B, n_neg=N, csr_indptr=_dev(ptr), csr_indices=_dev(idx),
                            coo_user=_dev(pairs[:, 0], torch.int32), coo_item=_dev(pairs[:, 1], torch.int32), seed=77,
                            lazy_items=False, prefetch=True, native=native, entry_cap=entry_cap)

    def run(sh, n, name_next_of_last=True):
        for s in range(n):
            nxt = ((s + 1) % n_slices) * G if (s + 1 < n or name_next_of_last) else None
            sh.step_sampled(order, (s % n_slices) * G, next_start=nxt)

    def raw(sh):   # the shards as they sit in memory (no flush: lazily updated rows included as they are)
        torch.cuda.synchronize()
        return [x.clone() for x in (sh.user_emb, sh.item_emb, sh.item_bias, sh.user_bias, sh.m['user_emb'], sh.v['user_emb'],
                                    sh.m['item_emb'], sh.v['item_emb'], sh.m['item_bias'], sh.v['item_bias'])]

    # calibration: kept entries per step on this rank with ample room -> a capacity the first steps fit and a later one does not
    sh = build(None)
    kept = []
    for s in range(40):
        sh.step_sampled(order, (s % n_slices) * G)
        kept.append(int(sh.last_batch()[0][-1].item()))
    sh.close()
    first_over = np.array([0], dtype=np.int64)
    cap = 0
    if rank == 0:
        k = np.array(kept)
        cap = int(k[:6].max())                       # steps 1..6 fit on rank 0 ...
        later = np.nonzero(k[6:] > cap)[0]
        assert len(later), 'calibration found no later, larger batch'
        first_over[0] = 6 + int(later[0])            # ... this (0-based) step is rank 0's first that does not
    t = torch.tensor([cap, int(first_over[0])], dtype=torch.int64)
    dist.broadcast(t, src=0)
    cap0, s_star = int(t[0]), int(t[1])
    cap_r = cap0 if rank == 0 else None              # the other ranks keep ample room: ONE rank overflows
    # run A: up to (not including) the overflowing step
    a = build(cap_r)
    run(a, s_star, name_next_of_last=False)      # (naming it would PREPARE the overflowing batch and flag the status word)
    a.check_status()
    before = raw(a)
    a.close()
    # run B: 300 steps; from the overflowing step on nothing may write a table, on ANY rank (the flag travels with the
    # host check only, so the other ranks' guard is what this rank's skipped exchange leaves them: their own steps go on
    # on stale but valid rows -- the job as a whole is dead and says so at the next check)
    b = build(cap_r)
    run(b, 300)
    after = raw(b)
    if rank == 0:
        for x, y in zip(before, after):
            assert torch.equal(x, y), 'a table changed after the first overflowing step'
    b.flush()                                        # the sweep is guarded too
    if rank == 0:
        for x, y in zip(before, raw(b)):
            assert torch.equal(x, y), 'the flush of a poisoned state wrote a table'
    raised = False
    try:
        b.check_status()
    except RuntimeError as e:
        raised = 'capacity' in str(e) or 'another rank' in str(e)
    assert raised, 'the overflow was not reported'
    b.close()
    open(os.path.join(out_dir, f'ok{rank}'), 'w').write(f'{cap0} {s_star}')
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('native', [False, 'host-staged'])
def test_capacity_overflow_never_updates_a_table(tmp_path, native):
    """entry_cap forced too small for ONE later batch of rank 0: 300 steps are issued without a host check; the rank's
    tables (parameters and moments, as they sit in memory) are bit-equal to their state before the first overflowing step,
    hsk_shard_flush included; check_status then raises on every rank.  (csrc/hsk_shard.inc: hsk_guard_skip.)"""
    mp.spawn(_overflow_worker, args=(2, _free_port(), str(tmp_path), native), nprocs=2, join=True)
    assert os.path.isfile(tmp_path / 'ok0') and os.path.isfile(tmp_path / 'ok1')


@pytest.mark.gpu
def test_cfg5_shaped_sharded_smoke(tmp_path):
    """BASELINE configs[4] row shape (D=1024, neg_train=200) with both tables sharded over two ranks; tables scaled
    to fit, catalogue large against the batch so the item shards run the lazy AdamW path as cfg5 would"""
    shape = dict(D=1024, N=200, U=400, I=30000, B=16)
    mp.spawn(_shard_worker, args=(2, _free_port(), str(tmp_path), 'gloo', shape, 6, 'auto', True), nprocs=2, join=True)
    _check_against_single_gpu(tmp_path, 2, shape, 6)


def _comm_worker(rank, world, port, out_dir):
    _init(rank, world, port, 'nccl')
    from hassaku_amd.dist import Comm
    comm = Comm()
    assert comm.native
    dev = torch.device('cuda', 0)
    a = torch.arange(12, dtype=torch.float32, device=dev).view(4, 3)
    out = torch.empty_like(a)
    w = comm.all_gather_into(out, a, async_op=True)
    busy = torch.ones(1 << 20, device=dev).cumsum(0)             # independent work queued under the collective
    w.wait()
    assert torch.equal(out, a)
    w = comm.reduce_scatter(out, a * 2, async_op=True)
    w.wait()
    assert torch.equal(out, a * 2)
    w = comm.all_to_all(out, a + 1, async_op=True)
    w.wait()
    assert torch.equal(out, a + 1)
    t = a.clone()
    comm.all_reduce(t)
    comm.all_reduce(t, op='max')
    assert torch.equal(t, a)
    assert torch.equal(comm.all_gather(a)[0], a)
    comm.broadcast(t, src=0)
    comm.barrier()
    assert float(busy[-1].item()) == float(1 << 20)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_comm_native_branches_on_rccl(tmp_path):
    mp.spawn(_comm_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)


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
    complete model.pth (user and item tables gathered from the shards) that evaluates to the same validation metrics."""
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
