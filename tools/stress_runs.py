#!/usr/bin/env python3
"""Randomised check that runs of steps issued from C (hsk_bprmf_train_steps: replayed HIP graphs of 64 steps with grouped
preparation at small batches, eager launches otherwise, the in-launch pipeline at large ones) train exactly like the same
steps issued one by one:     python tools/stress_runs.py [seconds] [seed]          (on a GPU box)
Random shapes (D of every alignment, batches 1 .. 2500, 1 .. 120 negatives), losses, optimisers, lazy / dense user and
item AdamW, uniform / popular sampling, a random split of the steps into runs: tables, moments, loss sums, bit for bit."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hassaku_amd import hip_ops as ops  # noqa: E402


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def csr_from_pairs(pairs, n_rows):
    key = np.unique(pairs[:, 0].astype(np.int64) * (1 << 32) + pairs[:, 1])
    rows, cols = key >> 32, key & 0xffffffff
    indptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(indptr, rows + 1, 1)
    return np.cumsum(indptr), cols.astype(np.int32)


def one_case(rng):
    D = int(rng.choice([6, 16, 30, 33, 64, 96, 128, 200, 256, 402, 512, 768, 1024]))
    B = int(rng.choice([1, 7, 32, 64, 128, 128, 256, 500, 1024, 1100, 2048, 2500]))
    N = int(rng.choice([1, 1, 3, 8, 9, 20, 50, 70, 120]))
    n_users = int(rng.randint(20, 1200))
    n_items = int(rng.randint(N + 40, 9000))
    while (n_users + n_items) * D > 2e7:
        n_users, n_items = max(20, n_users // 2), max(N + 40, n_items // 2)
    loss = str(rng.choice(['bpr', 'bpr', 'bce', 'sampled_softmax']))
    opt = str(rng.choice(['adamw', 'adamw', 'adam', 'adagrad']))
    lazy_u = bool(rng.rand() < 0.6)
    lazy_i = ['auto', True, False][int(rng.randint(3))]
    popular = rng.rand() < 0.25
    dens = min(0.4, float(rng.choice([8.0, 40.0, 150.0])) / n_items)
    pairs = np.argwhere(rng.rand(n_users, n_items) < dens)
    if len(pairs) < 2:
        return None, {}
    pairs = pairs[rng.permutation(len(pairs))]
    ptr, idx = csr_from_pairs(pairs, n_users)
    P = {'user_emb': (rng.randn(n_users, D) * 0.05).astype(np.float32),
         'item_emb': (rng.randn(n_items, D) * 0.05).astype(np.float32),
         'item_bias': (rng.randn(n_items) * 0.1).astype(np.float32)}
    n_total = int(rng.choice([5, 20, 70, 130, 200])) if B <= 256 else int(rng.randint(3, 12))
    n_pos = len(pairs)
    reps = -(-(n_total + 2) * B // n_pos)
    order = torch.from_numpy(np.concatenate([np.random.RandomState(int(rng.randint(1 << 30))).permutation(n_pos)
                                             for _ in range(reps)])).cuda()
    runs, s = [], 0
    while s < n_total:
        m = int(min(n_total - s, rng.choice([1, 2, 5, 64, 65, 100, 200])))
        runs.append(m)
        s += m
    alias = None
    if popular:
        pr, al = ops.build_alias_table(np.bincount(pairs[:, 1], minlength=n_items).astype(np.float64) ** 0.75 + 1e-3)
        alias = (dev(pr), dev(al))
    adj = float(np.log(n_items / N)) if loss == 'sampled_softmax' else 0.0
    desc = dict(D=D, B=B, N=N, n_users=n_users, n_items=n_items, loss=loss, opt=opt, lazy_users=lazy_u, lazy_items=lazy_i,
                popular=popular, runs=runs)
    res = []
    for chunked in (True, False):
        t = {k: dev(v) for k, v in P.items()}
        try:
            st = ops.BprMfFusedState(t['user_emb'], t['item_emb'], t['item_bias'], None, None, lr=1e-3, wd=1e-4, max_batch=B,
                                     max_cols=N + 1, seed=5, csr_indptr=dev(ptr), csr_indices=dev(idx),
                                     coo_user=dev(pairs[:, 0], torch.int32), coo_item=dev(pairs[:, 1], torch.int32),
                                     lazy_users=lazy_u, lazy_items=lazy_i, alias=alias, loss=loss, log_adjust=adj, optimizer=opt)
        except (ValueError, RuntimeError) as e:
            return None, dict(desc, refused=str(e)[:100])
        st.st.nnz = order.numel()
        s = 0
        try:
            if chunked:
                for m in runs:
                    if m == 1:
                        st.step_sampled(order, s * B, B, N)
                    else:
                        st.steps_sampled(order, s * B, m, B, N)
                    s += m
            else:
                for s in range(n_total):
                    st.step_sampled(order, s * B, B, N)
            st.flush()
            st.check_status()
        except RuntimeError as e:
            return False, dict(desc, error=('chunked ' if chunked else 'single ') + str(e)[:300])
        mom = {'m_' + k: v.cpu().numpy().copy() for k, v in st.m.items() if v is not None}
        mom.update({'v_' + k: v.cpu().numpy().copy() for k, v in st.v.items() if v is not None})
        res.append(({k: v.cpu().numpy().copy() for k, v in t.items()}, mom, st.pop_loss_sum(), st.graph_replays()))
        del st
    diff = []
    if res[0][2] != res[1][2]:
        diff.append(('loss_sum', res[0][2], res[1][2]))
    for k in res[0][0]:
        if not np.array_equal(res[0][0][k], res[1][0][k]):
            diff.append((k, int((res[0][0][k] != res[1][0][k]).sum())))
    for k in res[0][1]:
        if not np.array_equal(res[0][1][k], res[1][1][k]):
            diff.append((k, int((res[0][1][k] != res[1][1][k]).sum())))
    return (not diff), dict(desc, diff=diff, graph_replays=res[0][3])


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.RandomState(seed)
    t_end = time.time() + budget
    n = bad = refused = replays = 0
    while time.time() < t_end:
        ok, desc = one_case(rng)
        if ok is None:
            refused += 1
            continue
        n += 1
        replays += desc.get('graph_replays', 0)
        if not ok:
            bad += 1
            print('MISMATCH', desc, flush=True)
    print(f'{n} cases ({replays} replayed graphs), {refused} refused up front, {bad} mismatches', flush=True)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
