#!/usr/bin/env python3
"""Randomised check that the in-launch preparation pipeline (csrc/hsk_fused.hip: hsk_pipe_step; the riding sampler with its
LDS bitmap / staged search, the riding sort phases) trains exactly like the side-stream prefetch:
    python tools/stress_pipeline.py [seconds] [seed] [dims, e.g. 1024,2048]          (on a GPU box)
For random shapes inside the pipeline's range (D % 256 == 0, batch >= 2048, item table 4.5 .. 48 MB), random row lengths
(some users far beyond the prefetched 256 entries / the 1024-entry LDS row), uniform or popular sampling, lazy or dense
user AdamW and a random pattern of runs and hints: tables, moments, losses and the last batch, bit for bit."""
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


def one_case(rng, lib, dims=(256, 512, 768)):
    D = int(rng.choice(list(dims)))
    mb = float(rng.uniform(5.0, 40.0))
    n_items = int(mb * (1 << 20) / (4 * D))
    n_users = int(rng.randint(200, 1500))
    B = int(rng.choice([2048, 3072, 4096]))
    N = int(rng.choice([16, 17, 33, 64, 100]))
    popular = rng.rand() < 0.3
    loss = str(rng.choice(['bpr', 'bpr', 'bce', 'sampled_softmax']))      # (sampled softmax: no pipeline, both arms alike)
    opt = str(rng.choice(['adamw', 'adamw', 'adam', 'adagrad']))
    lazy = rng.rand() < 0.6
    per_user = rng.choice([20, 120, 400])
    dens = per_user / n_items
    rows = []
    for u in range(n_users):
        n = int(max(1, rng.poisson(per_user)))
        if u == 0:
            n = int(rng.choice([300, 1100, 2000]))          # beyond the prefetch registers / the LDS row
        rows.append(np.stack([np.full(n, u, dtype=np.int64), rng.randint(0, n_items, size=n)], axis=1))
    pairs = np.concatenate(rows)
    pairs = np.unique(pairs, axis=0)
    pairs = pairs[rng.permutation(len(pairs))]
    ptr, idx = csr_from_pairs(pairs, n_users)
    P = {'user_emb': (rng.randn(n_users, D) * 0.05).astype(np.float32),
         'item_emb': (rng.randn(n_items, D) * 0.05).astype(np.float32),
         'item_bias': (rng.randn(n_items) * 0.1).astype(np.float32)}
    n_pos = len(pairs)
    n_total = int(rng.randint(8, 16))
    reps = -(-(n_total + 4) * B // n_pos)
    order = torch.from_numpy(np.concatenate([np.random.RandomState(int(rng.randint(1 << 30))).permutation(n_pos)
                                             for _ in range(reps)])).cuda()
    # a random pattern of runs: (length, batches named behind it: 0, 1, 2, or a wrong guess)
    runs, s = [], 0
    while s < n_total:
        m = int(min(n_total - s, rng.randint(1, 6)))
        runs.append((m, int(rng.choice([0, 1, 2, 3]))))
        s += m
    alias = None
    if popular:
        pr, al = ops.build_alias_table(np.bincount(pairs[:, 1], minlength=n_items).astype(np.float64) ** 0.75 + 1e-3)
        alias = (dev(pr), dev(al))
    res = []
    for pipelined in (True, False):
        lib.hsk_bprmf_set_pipeline(1 if pipelined else 0)
        t = {k: dev(v.reshape(-1) if k == 'item_bias' else v) for k, v in P.items()}
        st = ops.BprMfFusedState(t['user_emb'], t['item_emb'], t['item_bias'], None, None, lr=1e-3, wd=1e-4, max_batch=B,
                                 max_cols=N + 1, seed=5, csr_indptr=dev(ptr), csr_indices=dev(idx),
                                 coo_user=dev(pairs[:, 0], torch.int32), coo_item=dev(pairs[:, 1], torch.int32),
                                 lazy_users=lazy, alias=alias, loss=loss, optimizer=opt,
                                 log_adjust=float(np.log(n_items / N)) if loss == 'sampled_softmax' else 0.0)
        st.st.nnz = order.numel()
        s = 0
        for m, hint in runs:
            nxt = s + m
            if hint == 1:
                st.hint_after_run(order, nxt * B, B, N, n_batches=1)
            elif hint == 2:
                st.hint_after_run(order, nxt * B, B, N, n_batches=2)
            elif hint == 3:
                st.hint_after_run(order, (nxt + 1) * B, B, N, n_batches=2)   # a wrong guess
            if m == 1 and hint == 0:
                st.step_sampled(order, s * B, B, N)
            else:
                st.steps_sampled(order, s * B, m, B, N)
            s = nxt
        st.flush()
        st.check_status()
        mom = {'m_' + k: v.cpu().numpy().copy() for k, v in st.m.items() if v is not None}
        mom.update({'v_' + k: v.cpu().numpy().copy() for k, v in st.v.items() if v is not None})
        bu, bi = st.last_batch(B, N + 1)
        res.append(({k: v.cpu().numpy().copy() for k, v in t.items()}, mom, st.pop_loss_sum(), bu.cpu().numpy(), bi.cpu().numpy(),
                    st.pipelined_steps()))
        del st
    diff = []
    if res[0][2] != res[1][2]:
        diff.append(('loss_sum', res[0][2], res[1][2]))
    if not np.array_equal(res[0][3], res[1][3]):
        diff.append(('last_batch_users', int((res[0][3] != res[1][3]).sum())))
    if not np.array_equal(res[0][4], res[1][4]):
        diff.append(('last_batch_items', int((res[0][4] != res[1][4]).sum())))
    for k in res[0][0]:
        if not np.array_equal(res[0][0][k], res[1][0][k]):
            d = np.abs(res[0][0][k].astype(np.float64) - res[1][0][k]).reshape(res[0][0][k].shape[0], -1).max(axis=1)
            diff.append((k, 'rows differing', int((d > 0).sum()), 'max abs', float(d.max())))
    for k in res[0][1]:
        if not np.array_equal(res[0][1][k], res[1][1][k]):
            diff.append((k, int((res[0][1][k] != res[1][1][k]).sum())))
    ok = not diff
    desc = dict(diff=diff, loss=loss, opt=opt, D=D, n_items=n_items, n_users=n_users, B=B, N=N, popular=popular, lazy=lazy, runs=runs, pipelined_steps=res[0][5])
    return ok, desc


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dims = tuple(int(x) for x in sys.argv[3].split(',')) if len(sys.argv) > 3 else (256, 512, 768)
    rng = np.random.RandomState(seed)
    lib = ops._lib.load()
    t_end = time.time() + budget
    n = bad = piped = 0
    try:
        while time.time() < t_end:
            ok, desc = one_case(rng, lib, dims)
            n += 1
            piped += desc['pipelined_steps']
            if not ok:
                bad += 1
                print('MISMATCH', desc, flush=True)
    finally:
        lib.hsk_bprmf_set_pipeline(1)
    print(f'{n} cases, {piped} pipelined steps, {bad} mismatches', flush=True)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
