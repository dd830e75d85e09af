#!/usr/bin/env python3
"""bench_eval.py -- full-catalogue evaluation throughput on one MI355X (users/s, effective fp32 TFLOP/s).

Not the driver's bench (that is bench.py); this measures the eval kernels of BASELINE configs 3 and 4:
scores = U_chunk x I^T (+bias, -inf on the exclude CSR), top-100, rank metrics.  Synthetic tables and CSRs.
    python bench_eval.py [--shape ml10m|lfm2b] [--chunk 2048] [--repeat 3]
Item-sharded over N GPUs (BASELINE configs[3]: every rank scores its I/N items, local top-100, all_gather of the
(value, id) candidates, merge, metrics on its share of the users):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench_eval.py --gpus N
(`--backend gloo` stages the candidate exchange through the host and lets the ranks share one GPU: rehearsal only.)
"""
import os
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
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--backend', default='nccl')
    args = ap.parse_args()
    U, I, D, npos = SHAPES[args.shape]
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    local = int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    comm = None
    if world > 1:
        import torch.distributed as dist
        from hassaku_amd.dist import Comm
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)
        comm = Comm()
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

    lo_i, hi_i = (I * rank) // world, (I * (rank + 1)) // world      # this rank's item shard

    def one_pass_sharded():
        acc = torch.zeros(3, dtype=torch.float64, device=dev)
        for lo in range(0, U, args.chunk):
            u = torch.arange(lo, min(lo + args.chunk, U), device=dev)
            v, i, _ = ops.mf_eval_topk(user_emb, item_emb, item_bias, None, None, u, 100, ep, ei, item_begin=lo_i,
                                       item_count=hi_i - lo_i, scores_ws=scores)
            cand_v = torch.stack(comm.all_gather(v.contiguous())).contiguous()      # [world, rows, 100]
            cand_i = torch.stack(comm.all_gather(i.contiguous())).contiguous()
            _, ids = ops.topk_merge(cand_v, cand_i)
            mine = slice(rank, len(u), world)                                        # metrics on this rank's share
            acc += ops.rank_metrics(ids[mine].contiguous(), u[mine].contiguous(), lp, li, ks)[:, 2].double().sum(0)
        comm.all_reduce(acc)
        return acc

    run = one_pass if world == 1 else one_pass_sharded
    ref = run()
    torch.cuda.synchronize()
    if comm is not None:
        comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.repeat):
        run()
    torch.cuda.synchronize()
    if comm is not None:
        comm.barrier()
    dt = (time.perf_counter() - t0) / args.repeat
    if comm is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({'shape': args.shape, 'U': U, 'I': I, 'D': D, 'chunk': args.chunk, 'n_gpus': world,
                          'sharding': 'items' if world > 1 else 'none', 'seconds_per_full_eval': dt,
                          'users_per_s': U / dt, 'tflops_fp32': 2.0 * U * I * D / dt / 1e12,
                          'ndcg_sum_check': [float(x) for x in ref.cpu()]}), flush=True)
    if comm is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
