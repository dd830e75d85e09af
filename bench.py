#!/usr/bin/env python3
"""bench.py -- BPR triplets/s of the fused MI355X BPR-MF training step (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ml10m|ml1m|ml100k]

One "step" = one full training step of Trainer.fit on one batch of synthetic interactions: on-device
uniform rejection sampling of the negatives, embedding gathers, (u.i - u.j) scores with item bias,
BPR log-sigmoid loss, gradients, and the AdamW update of every parameter row (dense semantics of
torch.optim.AdamW).  Inputs (tables, interaction CSR/COO, the epoch permutation) are resident in HBM
before the timed region.  Default workload: BASELINE.json configs[2] (ml10m shape, D=512, N=100,
B=4096) -- the configuration the metric "% HBM-read roofline at dim=512" is quoted on.

Rank 0 prints ONE JSON line (see the driver contract).  `roofline` describes the gather+BPR kernel
(k_fwd_ugrad): algorithmic read bytes per launch / its mean duration, measured with HIP events on
the launch stream inside the timed region.  `cpu_baseline` times the CPU restatement of the
reference's trainer (oracle/cpu_trainer.py, kind "port") on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

WORKLOADS = {
    # name: (synthetic shape, D, n_neg, batch)  -- BASELINE.json configs[0..2]
    'ml100k': ('ml100k', 64, 1, 128),
    'ml1m': ('ml1m', 402, 50, 128),
    'ml10m': ('ml10m', 512, 100, 4096),
    'lfm2b': ('lfm2b', 512, 100, 4096),   # the BASELINE configs[3] catalogue (131 072 items) under the configs[2] step
    # not a BASELINE config: the ml10m step on tables that cannot be cached (the honest HBM point for k_fwd_ugrad)
    'hbm': ('hbm', 512, 100, 4096),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILE_DIR = 'r1e_final'  # committed rocprofv3 summaries of `bench.py` (profiles/README.md)
LR, WD = 3e-4, 4e-5    # README.md:82-83 of the reference (canonical BPR-MF conf)


def fwd_read_bytes(B, N, D):
    """Algorithmic HBM read bytes of one k_fwd_ugrad launch (negatives read from memory):
    user rows + (1+N) item rows per positive, item bias, item ids, user ids."""
    return 4 * D * B * (2 + N) + 4 * B * (1 + N) + 4 * B * (1 + N) + 4 * B


def pmc_traffic_bytes(kernel_prefix, profile_dir):
    """HBM-side traffic of one launch from the committed rocprofv3 PMC passes of this same command
    (`tools_profile.sh`; FETCH_SIZE and WRITE_SIZE are collected in separate runs and reported in KB).  On gfx950
    FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads (MI355X_MICROARCH.md, HBM): x2."""
    path = os.path.join(REPO, 'profiles', profile_dir, 'pmc_summary.json')
    if not os.path.isfile(path):
        return None
    for name, row in json.load(open(path)).items():
        if name.startswith(kernel_prefix) and 'FETCH_SIZE_KB_mean' in row and 'WRITE_SIZE_KB_mean' in row:
            return (2.0 * row['FETCH_SIZE_KB_mean'] + row['WRITE_SIZE_KB_mean']) * 1024.0
    return None


def build_state(data, D, B, N, device, seed=64, **kw):
    from hassaku_amd import hip_ops as ops
    from hassaku_amd.data.csr import UserItemCsr
    U, I = data.n_users, data.n_items
    csr = UserItemCsr.from_pairs(data.train[:, 0], data.train[:, 1], U, I)
    torch.manual_seed(seed)
    user_emb = torch.empty((U, D), device=device).normal_(std=0.1 / D)   # train/utils.py:12-13 of the reference
    item_emb = torch.empty((I, D), device=device).normal_(std=0.1 / D)
    item_bias = torch.empty((I,), device=device).normal_(std=0.1)        # [I,1] table: std 0.1/1
    indptr, indices = csr.to_device(device)
    coo_u = torch.from_numpy(data.train[:, 0].astype(np.int32)).to(device)
    coo_i = torch.from_numpy(data.train[:, 1].astype(np.int32)).to(device)
    st = ops.BprMfFusedState(user_emb, item_emb, item_bias, lr=LR, wd=WD, max_batch=B, max_cols=N + 1, seed=seed,
                             csr_indptr=indptr, csr_indices=indices, coo_user=coo_u, coo_item=coo_i, **kw)
    return st, csr


def build_sharded_state(data, D, B, N, device, seed=64):
    """world > 1: row-sharded user tables + replicated item table (hassaku_amd/dist.py); B positives per rank."""
    from hassaku_amd.data.csr import UserItemCsr
    from hassaku_amd.dist import Comm, ShardedBprMf
    comm = Comm()
    U, I = data.n_users, data.n_items
    csr = UserItemCsr.from_pairs(data.train[:, 0], data.train[:, 1], U, I)
    torch.manual_seed(seed)                                   # same initial tables on every rank
    user_emb = torch.empty((U, D), device=device).normal_(std=0.1 / D)
    item_emb = torch.empty((I, D), device=device).normal_(std=0.1 / D)
    item_bias = torch.empty((I,), device=device).normal_(std=0.1)
    indptr, indices = csr.to_device(device)
    coo_u = torch.from_numpy(data.train[:, 0].astype(np.int32)).to(device)
    coo_i = torch.from_numpy(data.train[:, 1].astype(np.int32)).to(device)
    st = ShardedBprMf(comm, user_emb, item_emb, item_bias, lr=LR, wd=WD, batch=B, n_neg=N, csr_indptr=indptr,
                      csr_indices=indices, coo_user=coo_u, coo_item=coo_i, seed=seed)
    del user_emb
    return st, csr, comm


def hip_stage_names(st):
    from hassaku_amd.hip_ops import BprMfFusedState
    return BprMfFusedState.STAGES if isinstance(st, BprMfFusedState) else ('fwd', 'item')


def usable_cores():
    """CPU share of this process: affinity mask, capped by the cgroup quota (a GPU box hands out 16 of its cores)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    if n > 64:      # no quota visible: stay within the documented per-GPU share
        n = 16
    return n


def cpu_baseline(data, csr, D, N, B, budget_s):
    from oracle.cpu_trainer import CpuTrainer
    cores = usable_cores()
    tr = CpuTrainer(data.n_users, data.n_items, D, LR, WD, csr.indptr, csr.indices, data.train[:, 0], data.train[:, 1],
                    N, B, threads=cores)
    steps, secs = tr.time_steps(budget_s=budget_s)
    return {'value': steps * B * N / secs, 'unit': 'triplets/s', 'cores': cores, 'kind': 'port',
            'sample': f'{steps} steps of B={B} x N={N} (D={D}) in {secs:.1f}s, torch CPU + numpy sampler, 0 workers'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='ml10m', choices=sorted(WORKLOADS))
    ap.add_argument('--cpu-budget', type=float, default=15.0, help='seconds of CPU-baseline work (0 = skip)')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for --gpus > 1 ('nccl' = RCCL; 'gloo' "
                    "stages collectives through the host and lets several ranks share one GPU: functional rehearsal only)")
    ap.add_argument('--no-prefetch', action='store_true', help='do not prepare (sample + sort) the next batch on a side stream during the current step')
    ap.add_argument('--dense-users', action='store_true', help='dense AdamW sweep over the user table every step')
    ap.add_argument('--time-all-stages', action='store_true', help='event-time every stage (perturbs the step time)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (no CPU fallback)')
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(args.backend)

    from hassaku_amd.data import synthetic
    shape, D, N, B = WORKLOADS[args.workload]
    data = synthetic.generate_named(shape, seed=0)          # same seed on every rank: identical data everywhere
    nnz = data.train.shape[0]
    comm = None
    if world == 1:
        st, csr = build_state(data, D, B, N, device, overlap=not args.no_prefetch, lazy_users=not args.dense_users)
    else:
        st, csr, comm = build_sharded_state(data, D, B, N, device)

    gen = torch.Generator(device=device)
    gen.manual_seed(64)
    order = torch.randperm(nnz, device=device, generator=gen)
    if comm is not None:
        comm.broadcast(order, src=0)                              # one epoch order for the whole job
    n_batches = nnz // (B * world)

    def run(n, first):
        if world == 1:
            # the epoch's inner loop, issued from C in runs of consecutive batches (hsk_bprmf_train_steps: each step
            # hints the next one to the prefetch); a run ends where the epoch order wraps around
            s = 0
            while s < n:
                k0 = (first + s) % n_batches
                m = min(n - s, n_batches - k0, 256)
                st.steps_sampled(order, k0 * B, m, B, N)
                s += m
            return
        for s in range(n):
            start = ((first + s) % n_batches) * B * world        # global batch = world * B positives (weak scaling)
            st.step_sampled(order, start)

    def fence():
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    run(args.warmup, 0)
    fence()
    st.check_status('warm-up')
    # every stage on every step when asked; otherwise only the roofline kernel, on every 8th step, so that the
    # event records (each costs a few us of launch gap) do not distort the step time being measured
    stages = hip_stage_names(st) if args.time_all_stages else ('fwd',)
    st.enable_timing(stages, every=1 if args.time_all_stages else 8)
    fence()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    st.flush()   # lazily updated user rows are brought up to date INSIDE the timed region: no work is skipped
    fence()
    t1 = time.perf_counter()
    st.disable_timing()
    elapsed = t1 - t0
    if comm is not None:                                          # the slowest rank defines the step time
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timing = st.collect_timing()
    st.check_status('timed region')
    loss = st.last_loss()
    assert np.isfinite(loss), loss

    ms_per_step = elapsed * 1e3 / args.steps
    value = args.steps * B * N * world / elapsed
    fwd_ms, fwd_n = timing['fwd']
    fwd_us = fwd_ms * 1e3 / fwd_n
    achieved = fwd_read_bytes(B, N, D) / (fwd_us * 1e-6) / 1e9
    out = {
        'metric': 'BPR triplets/sec', 'value': value, 'unit': 'triplets/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'{args.workload}-shaped synthetic, mf + bpr + adamw, embedding_dim={D}, '
                               f'neg_train={N}, batch={B}, U={data.n_users}, I={data.n_items}, nnz_train={nnz}',
                   'global_batch': B * world,
                   'parallelism': 'single GPU' if world == 1 else
                   f'{world} ranks: user tables row-sharded (all_to_all), item table replicated (all_reduce)',
                   'lr': LR, 'wd': WD,
                   'loss_last_step': loss},
        'roofline': {'bound': 'hbm', 'kernel': 'k_fwd_ugrad (gather + scores + BPR + user-row grad)',
                     'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                     'traffic': pmc_traffic_bytes('k_fwd_ugrad', PROFILE_DIR) if args.workload == 'ml10m' else None,
                     'traffic_source': f'profiles/{PROFILE_DIR}/pmc_summary.json (rocprofv3 --pmc, same command)',
                     'avg_us': fwd_us, 'launches': fwd_n,
                     'algorithmic_bytes_per_launch': fwd_read_bytes(B, N, D)},
    }
    if args.time_all_stages:
        out['stage_us_per_step'] = {k: v[0] * 1e3 / args.steps for k, v in timing.items()}
    if rank == 0 and world == 1 and args.cpu_budget > 0:
        out['cpu_baseline'] = cpu_baseline(data, csr, D, N, B, args.cpu_budget)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
