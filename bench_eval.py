#!/usr/bin/env python3
"""bench_eval.py -- full-catalogue evaluation throughput on one MI355X (users/s, effective fp32 TFLOP/s).

Not the driver's bench (that is bench.py); this measures the eval kernels of BASELINE configs 3 and 4:
scores = U_chunk x I^T (+bias, -inf on the exclude CSR), top-100, rank metrics.  Synthetic tables and CSRs.
    python bench_eval.py [--shape ml10m|lfm2b] [--chunk 2048] [--repeat 3]
"""
import argparse
import json
import time

import numpy as np
import torch

from hassaku_amd import hip_ops as ops
from hassaku_amd.data.csr import UserItemCsr

SHAPES = {'ml10m': (69878, 10677, 512, 82), 'lfm2b': (16384, 131072, 512, 120)}   # U, I, D, positives per user


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='ml10m', choices=sorted(SHAPES))
    ap.add_argument('--chunk', type=int, default=2048)
    ap.add_argument('--repeat', type=int, default=3)
    args = ap.parse_args()
    U, I, D, npos = SHAPES[args.shape]
    dev = torch.device('cuda')
    torch.manual_seed(0)
    user_emb = torch.randn(U, D, device=dev) * 0.05
    item_emb = torch.randn(I, D, device=dev) * 0.05
    item_bias = torch.randn(I, device=dev) * 0.1
    rng = np.random.default_rng(0)
    users = np.repeat(np.arange(U), npos)
    excl = UserItemCsr.from_pairs(users, rng.integers(0, I, size=len(users)), U, I)
    lab = UserItemCsr.from_pairs(np.repeat(np.arange(U), 10), rng.integers(0, I, size=10 * U), U, I)
    ep, ei = excl.to_device(dev)
    lp, li = lab.to_device(dev)
    ks = [100, 50, 10, 5]
    scores = torch.empty((args.chunk, I), dtype=torch.float32, device=dev)

    def one_pass():
        acc = torch.zeros(3, dtype=torch.float64, device=dev)
        for lo in range(0, U, args.chunk):
            u = torch.arange(lo, min(lo + args.chunk, U), device=dev)
            _, ids, _ = ops.mf_eval_topk(user_emb, item_emb, item_bias, None, None, u, 100, ep, ei, scores_ws=scores)
            acc += ops.rank_metrics(ids, u, lp, li, ks)[:, 2].double().sum(0)
        return acc

    one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.repeat):
        one_pass()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.repeat
    print(json.dumps({'shape': args.shape, 'U': U, 'I': I, 'D': D, 'chunk': args.chunk, 'seconds_per_full_eval': dt,
                      'users_per_s': U / dt, 'tflops_fp32': 2.0 * U * I * D / dt / 1e12}))


if __name__ == '__main__':
    main()
