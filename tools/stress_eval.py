#!/usr/bin/env python3
"""Randomised cross-check of the evaluation paths (run on a GPU box: python tools/stress_eval.py [seconds] [seed]).
For random shapes (rows, users, items, D, k, item range, exclusions, biases) and every arithmetic form:
  * fused top-k (no score matrix) == materialised top-k, values / ids / order, bit for bit;
  * materialised scores against float64 within 4e-6 of the largest score;
  * in 30 % of the cases: the item range handed over as a physical shard (item_shard=True) gives the same top-k.
Prints one line per case that fails and a summary; exit code 1 on any failure."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from hassaku_amd import hip_ops as ops  # noqa: E402


def csr_from_pairs(pairs, n_users):
    order = np.lexsort((pairs[:, 1], pairs[:, 0]))
    pairs = pairs[order]
    ptr = np.zeros(n_users + 1, dtype=np.int64)
    np.add.at(ptr, pairs[:, 0] + 1, 1)
    return np.cumsum(ptr), pairs[:, 1].astype(np.int32)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.RandomState(seed)
    t_end = time.time() + budget
    n_case = n_fail = 0
    while time.time() < t_end:
        n_users = int(rng.randint(1, 900))
        n_items = int(rng.choice([rng.randint(1, 300), rng.randint(300, 5000), rng.randint(5000, 60000)]))
        D = int(rng.choice([4, 8, 16, 32, 36, 64, 100, 128, 200, 256, 402, 512]))
        R = int(rng.randint(1, 700))
        lo = int(rng.randint(0, n_items)) if rng.rand() < 0.5 else 0
        cnt = int(rng.randint(1, n_items - lo + 1))
        k = int(min(cnt, rng.choice([1, 5, 10, 50, 100, 128])))
        scale = float(10.0 ** rng.uniform(-4, 2))
        g = torch.Generator(device='cuda').manual_seed(int(rng.randint(1 << 30)))
        U = torch.randn(n_users, D, device='cuda', generator=g) * scale
        I = torch.randn(n_items, D, device='cuda', generator=g) * scale
        if n_items > 40 and rng.rand() < 0.5:
            I[n_items // 2: n_items // 2 + 10] = I[:10]          # exact ties
        Ib = (torch.randn(n_items, device='cuda', generator=g) * scale * scale) if rng.rand() < 0.7 else None
        Ub = (torch.randn(n_users, device='cuda', generator=g) * scale * scale) if rng.rand() < 0.3 else None
        gb = torch.tensor([0.1 * scale * scale], device='cuda') if rng.rand() < 0.3 else None
        e_ptr = e_idx = None
        if rng.rand() < 0.7:
            dens = min(0.5, float(rng.choice([5.0, 50.0, 300.0])) / n_items)
            pairs = np.argwhere(rng.rand(n_users, n_items) < dens)
            if len(pairs):
                p, i = csr_from_pairs(pairs, n_users)
                e_ptr, e_idx = torch.from_numpy(p).cuda(), torch.from_numpy(i).cuda()
        u = torch.from_numpy(rng.randint(0, n_users, size=R).astype(np.int64)).cuda()
        shard = bool(rng.rand() < 0.3)
        what = dict(shard=shard, n_users=n_users, n_items=n_items, D=D, R=R, lo=lo, cnt=cnt, k=k, scale=scale, Ib=Ib is not None,
                    Ub=Ub is not None, gb=gb is not None, excl=e_ptr is not None)
        for form in (ops.EVAL_ARITH_F16X2, ops.EVAL_ARITH_BF16X3, ops.EVAL_ARITH_FP32):
            ops.set_eval_arith(form)
            v0, i0, sc = ops.mf_eval_topk(U, I, Ib, Ub, gb, u, k, e_ptr, e_idx, item_begin=lo, item_count=cnt, want_scores=True)
            v1, i1, _ = ops.mf_eval_topk(U, I, Ib, Ub, gb, u, k, e_ptr, e_idx, item_begin=lo, item_count=cnt, want_scores=False)
            n_case += 1
            ok = torch.equal(v0.view(torch.int32), v1.view(torch.int32)) and torch.equal(i0, i1)
            if shard:   # the same range handed over as a PHYSICAL shard (the library gets a virtual base and must never
                # read outside the shard's rows): same top-k, materialised and fused
                Is, Ibs = I[lo:lo + cnt].contiguous(), (None if Ib is None else Ib[lo:lo + cnt].contiguous())
                for want in (True, False):
                    v2, i2, _ = ops.mf_eval_topk(U, Is, Ibs, Ub, gb, u, k, e_ptr, e_idx, item_begin=lo, item_count=cnt,
                                                 item_shard=True, n_items_global=n_items, want_scores=want)
                    ok = ok and torch.equal(v0.view(torch.int32), v2.view(torch.int32)) and torch.equal(i0, i2)
            ref = U[u].double() @ I[lo:lo + cnt].double().T
            if Ub is not None:
                ref += Ub[u].double()[:, None]
            if Ib is not None:
                ref += Ib[lo:lo + cnt].double()[None, :]
            if gb is not None:
                ref += gb.double()
            got = sc[:R * cnt].view(R, cnt).double()
            fin = torch.isfinite(got)
            err = ((got - ref)[fin].abs().max().item() if fin.any() else 0.0)
            big = max(ref.abs().max().item(), 1e-30)
            ok_acc = err <= 4e-6 * big
            if not (ok and ok_acc):
                n_fail += 1
                print('FAIL', 'form', form, 'paths_equal', ok, 'err/max', err / big, what, flush=True)
    ops.set_eval_arith(ops.EVAL_ARITH_DEFAULT)
    print(f'{n_case} cases, {n_fail} failures', flush=True)
    sys.exit(1 if n_fail else 0)


if __name__ == '__main__':
    main()
