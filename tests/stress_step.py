#!/usr/bin/env python3
"""Randomised parity of the fused training step against the oracle (test infrastructure; not collected by pytest):
    python tests/stress_step.py [seconds] [seed] [big]          (on a GPU box; big: only the partitioned forward's shapes)
Random shapes (D 2 .. 2048 of every alignment, batches 1 .. 4500, 1 .. 200 negatives, random table sizes), losses
(bpr / bce / sampled_softmax), optimisers (adamw / adam / adagrad), bias sets, lazy or dense AdamW, duplicate users /
items in a batch; 2-3 steps on caller-given batches, then loss, parameters and moments against oracle.MfOracleTrainer
under the tolerance rules of tests/conftest.py (assert_adam_param_close)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import assert_adam_param_close  # noqa: E402
from hassaku_amd import hip_ops as ops  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def one_case(rng, big=False):
    D = int(rng.choice([2, 6, 16, 30, 33, 64, 100, 128, 200, 256, 384, 402, 512, 640, 768, 1024, 1280, 2048]))
    B = int(rng.choice([1, 5, 17, 64, 128, 300, 1030, 2048, 2100, 4096, 4500]))
    if big:   # the item-partitioned forward's range: whole-chunk rows, batches >= 2048, a 4.5 .. 48 MB item table
        D = int(rng.choice([256, 512, 1024, 2048]))
        B = int(rng.choice([2048, 2100, 3000, 4096]))
    N = int(rng.choice([1, 3, 8, 9, 20, 50, 100, 200]))
    while B * (N + 1) * D > 5e8:                  # keep the oracle's dense step in seconds
        B = max(1, B // 2)
    U = int(rng.randint(max(2, B // 50), 3000))
    I = int(rng.randint(max(N + 2, 250), 20000))   # (the rule's 0.5 % of elements needs a few hundred of them)
    if big:
        I = int(rng.uniform(5.0, 40.0) * (1 << 20) / (4 * D))
    while (U + I) * D > 6e7 and not big:
        U, I = max(2, U // 2), max(N + 2, I // 2)
    loss = str(rng.choice(['bpr', 'bpr', 'bce', 'sampled_softmax']))
    opt = str(rng.choice(['adamw', 'adamw', 'adam', 'adagrad']))
    with_ib = rng.rand() < 0.8
    with_ub = rng.rand() < 0.3
    with_gb = rng.rand() < 0.3
    lazy = bool(rng.rand() < 0.5)
    lr, wd = float(10 ** rng.uniform(-4, -2.5)), float(rng.choice([0.0, 4e-5, 1e-3]))   # (beyond 3e-3 Adam turns the rounding
    # noise of near-zero gradients into parameter differences past the tolerance rule: not what this looks for)
    P = {'user_emb': (rng.randn(U, D) * 0.1).astype(np.float32), 'item_emb': (rng.randn(I, D) * 0.1).astype(np.float32)}
    if with_ib:
        P['item_bias'] = (rng.randn(I) * 0.1).astype(np.float32)
    if with_ub:
        P['user_bias'] = (rng.randn(U) * 0.1).astype(np.float32)
    if with_gb:
        P['global_bias'] = (rng.randn(1) * 0.1).astype(np.float32)
    desc = dict(D=D, B=B, N=N, U=U, I=I, loss=loss, opt=opt, biases=(with_ib, with_ub, with_gb), lazy=lazy, lr=lr, wd=wd)
    t = {k: dev(v) for k, v in P.items()}
    log_adjust = float(np.log(I / N)) if loss == 'sampled_softmax' else 0.0
    try:
        st = ops.BprMfFusedState(t['user_emb'], t['item_emb'], t.get('item_bias'), t.get('user_bias'), t.get('global_bias'),
                                 lr=lr, wd=wd, max_batch=B, max_cols=N + 1, loss=loss, log_adjust=log_adjust, optimizer=opt,
                                 lazy_users=lazy)
    except (ValueError, RuntimeError) as e:       # a combination the library refuses up front is not a parity case
        return None, dict(desc, refused=str(e)[:120])
    tr = orc.MfOracleTrainer(P['user_emb'], P['item_emb'], P.get('item_bias'), P.get('user_bias'), P.get('global_bias'), lr=lr,
                             wd=wd, loss=loss, log_adjust=log_adjust, optimizer=opt)
    try:
        for _ in range(int(rng.randint(2, 4))):
            u = rng.randint(0, U, size=B).astype(np.int64)
            i = rng.randint(0, I, size=(B, N + 1)).astype(np.int64)
            if B > 3 and rng.rand() < 0.5:
                u[: max(2, min(B // 10, 64))] = u[0]                       # one user several times
            if rng.rand() < 0.3:
                i[: max(1, min(B // 10, 64)), 1:] = i[0, 1]                # one negative in whole rows
            st.step(dev(u), dev(i))
            loss_ref = tr.step(u, i)[0]
            got = st.last_loss()
            assert abs(got - loss_ref) <= 2e-6 * max(abs(loss_ref), 1e-3), ('loss', got, loss_ref)
        st.flush()
        st.check_status()
        tol = dict(max_tol=5e-3, frac=None) if opt == 'adagrad' else {}
        for name in P:
            if st.m.get(name) is not None:
                assert_adam_param_close(st.m[name].cpu().numpy(), tr.M[name], ('m', name), **tol)
                assert_adam_param_close(st.v[name].cpu().numpy(), tr.V[name], ('v', name), **tol)
            assert_adam_param_close(t[name].cpu().numpy(), tr.P[name], name, **tol)
    except AssertionError as e:
        return False, dict(desc, error=str(e)[:300])
    except RuntimeError as e:
        return False, dict(desc, error='RuntimeError: ' + str(e)[:300])
    return True, desc


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    big = len(sys.argv) > 3 and sys.argv[3] == 'big'
    rng = np.random.RandomState(seed)
    orc.build()
    t_end = time.time() + budget
    n = bad = refused = 0
    while time.time() < t_end:
        ok, desc = one_case(rng, big)
        if ok is None:
            refused += 1
            if refused <= 5:
                print('refused', desc, flush=True)
            continue
        n += 1
        if not ok:
            bad += 1
            print('FAIL', desc, flush=True)
    print(f'{n} cases, {refused} refused up front, {bad} failures', flush=True)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
