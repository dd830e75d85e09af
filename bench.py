#!/usr/bin/env python3
"""bench.py -- BPR triplets/s of the fused MI355X BPR-MF training step (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ml10m|ml1m|ml100k|lfm2b|hbm]

One "step" = one full training step of Trainer.fit on one batch of synthetic interactions: on-device
uniform rejection sampling of the negatives, embedding gathers, (u.i - u.j) scores with item bias,
BPR log-sigmoid loss, gradients, and the AdamW update of every parameter row (dense semantics of
torch.optim.AdamW).  Inputs (tables, interaction CSR/COO, the epoch permutation) are resident in HBM
before the timed region.  Default workload: BASELINE.json configs[2] (ml10m shape, D=512, N=100,
B=4096) -- the configuration the metric "% HBM-read roofline at dim=512" is quoted on.

Rank 0 prints ONE JSON line (see the driver contract).  Besides the contract's keys it carries, all timed inside
this run:
  roofline      the gather+BPR kernel (k_fwd_ugrad) of the headline workload: algorithmic read bytes per launch / its
                mean duration (HIP events of the kernel's own dispatch on the launch stream, inside the timed region).
                At the ml10m shape the tables live in L2 + Infinity Cache, so the bound is the cache fabric, not HBM
  workloads     (N=1) the other BASELINE training configs -- ml1m (configs[1]), ml100k (configs[0]) -- and `hbm`,
                the same step on tables no cache can hold, whose roofline IS the HBM-read roofline of the kernel
  eval          full-catalogue evaluation users/s (ml10m and lfm2b shapes at N=1; item-sharded lfm2b at N>1)
  cpu_baseline  (N=1) the CPU restatement of the reference's trainer (oracle/cpu_trainer.py, kind "port"): a bounded
                sample of the headline workload + the SURVEY 8(d) protocol on configs[0] (20 + 200 steps, 0 and 4
                loader workers)
N > 1 (one process per GPU, RCCL): item table range-sharded, user table row-sharded, global batch N*B (weak scaling).
"""
import os

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC for RCCL; must be set before HIP initialises
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')

import argparse  # noqa: E402
import json      # noqa: E402
import sys       # noqa: E402
import time      # noqa: E402

import numpy as np  # noqa: E402
import torch        # noqa: E402

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
EVAL_SHAPES = {'ml10m': (69878, 10677, 512, 82), 'lfm2b': (16384, 131072, 512, 120)}   # U, I, D, positives per user
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured streaming copy)
ICACHE_GATHER_GBS = 8600.0   # same guide, "Indexed rows": uniformly random rows of an Infinity-Cache-resident table
L2_GATHER_GBS = 17800.0      # same table: rows shared by every workgroup of an XCD (its L2), 16.8-18.8 TB/s chip-wide
HBM_GATHER_GBS = 5750.0      # same guide: random 2.3 KB rows of a table far beyond the Infinity Cache, 5.7-5.8 TB/s
MFMA_FP32_TFLOPS = 157.0     # exact-fp32 MFMA peak (v_mfma_f32_32x32x2_f32; no xf32 on gfx950)
MFMA_BF16_TFLOPS = 2500.0    # dense bf16 MFMA peak (MI355X_MICROARCH.md)
PROFILE_DIR = {'ml10m': 'r4c_ml10m', 'hbm': 'r4c_hbm', 'ml1m': 'r4_ml1m', 'ml100k': 'r4_ml100k'}  # committed rocprofv3 summaries (PMC passes)
LR, WD = 3e-4, 4e-5          # README.md:82-83 of the reference (canonical BPR-MF conf)


def fwd_read_bytes(B, N, D):
    """Algorithmic HBM read bytes of one k_fwd_ugrad launch (negatives read from memory):
    user rows + (1+N) item rows per positive, item bias, item ids, user ids."""
    return 4 * D * B * (2 + N) + 4 * B * (1 + N) + 4 * B * (1 + N) + 4 * B


def pmc_traffic(kernel_prefix, workload):
    """HBM-side traffic of one launch from the COMMITTED rocprofv3 PMC passes of this same command (tools/profile.sh;
    FETCH_SIZE and WRITE_SIZE are collected in separate runs and reported in KB).  On gfx950 FETCH_SIZE counts 64 B
    per 128-B request for wide coalesced reads (MI355X_MICROARCH.md, HBM): x2.  -> (bytes | None, source)."""
    d = PROFILE_DIR.get(workload)
    path = os.path.join(REPO, 'profiles', d or '', 'pmc_summary.json')
    if not d or not os.path.isfile(path):
        return None, None
    for name, row in json.load(open(path)).items():
        if name.startswith(kernel_prefix) and 'FETCH_SIZE_KB_mean' in row and 'WRITE_SIZE_KB_mean' in row:
            return ((2.0 * row['FETCH_SIZE_KB_mean'] + row['WRITE_SIZE_KB_mean']) * 1024.0,
                    f'committed profile profiles/{d}/pmc_summary.json (rocprofv3 --pmc of this command; not re-measured by this run)')
    return None, None


def pmc_fetch_bytes(kernel_prefix, workload, key=None):
    """Fabric-side READ bytes of one launch (FETCH_SIZE x 2, see pmc_traffic) from the committed profile, or None;
    key: another field of that kernel's row instead (e.g. 'l2_hit_rate' = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum))."""
    d = PROFILE_DIR.get(workload)
    path = os.path.join(REPO, 'profiles', d or '', 'pmc_summary.json')
    if not d or not os.path.isfile(path):
        return None
    for name, row in json.load(open(path)).items():
        if name.startswith(kernel_prefix) and 'FETCH_SIZE_KB_mean' in row:
            return row.get(key) if key else 2.0 * row['FETCH_SIZE_KB_mean'] * 1024.0
    return None


def init_tables(U, I, D, device, seed=64):
    torch.manual_seed(seed)
    user_emb = torch.empty((U, D), device=device).normal_(std=0.1 / D)   # train/utils.py:12-13 of the reference
    item_emb = torch.empty((I, D), device=device).normal_(std=0.1 / D)
    item_bias = torch.empty((I,), device=device).normal_(std=0.1)        # [I,1] table: std 0.1/1
    return user_emb, item_emb, item_bias


def device_interactions(data, device):
    from hassaku_amd.data.csr import UserItemCsr
    csr = UserItemCsr.from_pairs(data.train[:, 0], data.train[:, 1], data.n_users, data.n_items)
    indptr, indices = csr.to_device(device)
    coo_u = torch.from_numpy(data.train[:, 0].astype(np.int32)).to(device)
    coo_i = torch.from_numpy(data.train[:, 1].astype(np.int32)).to(device)
    return csr, dict(csr_indptr=indptr, csr_indices=indices, coo_user=coo_u, coo_item=coo_i)


def build_state(data, D, B, N, device, seed=64, **kw):
    from hassaku_amd import hip_ops as ops
    csr, arrays = device_interactions(data, device)
    user_emb, item_emb, item_bias = init_tables(data.n_users, data.n_items, D, device, seed)
    st = ops.BprMfFusedState(user_emb, item_emb, item_bias, lr=LR, wd=WD, max_batch=B, max_cols=N + 1, seed=seed,
                             **arrays, **kw)
    return st, csr


def build_sharded_state(data, D, B, N, device, comm, seed=64, native='auto', arrays=None):
    """world > 1: item table range-sharded, user table row-sharded (hassaku_amd/dist.py); B positives per rank."""
    from hassaku_amd.dist import ShardedBprMf
    csr = None
    if arrays is None:
        csr, arrays = device_interactions(data, device)
    user_emb, item_emb, item_bias = init_tables(data.n_users, data.n_items, D, device, seed)   # same on every rank
    st = ShardedBprMf(comm, user_emb, item_emb, item_bias, lr=LR, wd=WD, batch=B, n_neg=N, seed=seed, native=native,
                      **arrays)
    del user_emb, item_emb, item_bias
    return st, csr


def native_step_self_check(data, D, B, N, device, comm, order, n_steps=3):
    """Before a multi-rank job is timed: the natively issued step (hsk_shard_step, the library's own RCCL communicator,
    collectives from C on two streams) against the phase-by-phase step (torch.distributed between the same kernels), from
    the same initial state, on the job's own ranks and links.  -> dict(path, verdict, max_abs_diff).  'bit-equal' or
    'equal within 1e-6 (reduction order)' lets the native path be timed; anything else fails the run on every rank."""
    from hassaku_amd import _lib
    if not (comm.native and _lib.load().hsk_rccl_available()):
        return {'path': 'phased (torch.distributed)', 'verdict': 'native path not available', 'max_abs_diff': None}
    csr, arrays = device_interactions(data, device)
    G = B * comm.world
    shards, losses = [], []
    for native in (True, False):
        st, _ = build_sharded_state(data, D, B, N, device, comm, native=native, arrays=arrays)
        if native and not st.issued_natively:
            return {'path': 'phased (torch.distributed)', 'verdict': 'hsk_shard_rt_create failed on a rank', 'max_abs_diff': None}
        ls = []
        for s in range(n_steps):
            st.step_sampled(order, s * G, next_start=(s + 1) * G)
            ls.append(st.last_loss())
        st.flush()
        st.check_status('native / phased self-check')
        torch.cuda.synchronize()
        shards.append([t.clone() for t in (st.user_emb, st.item_emb, st.item_bias, st.m['user_emb'], st.v['user_emb'],
                                            st.m['item_emb'], st.v['item_emb'])])
        losses.append(ls)
        st.close()
        del st
        torch.cuda.empty_cache()
    bit = all(torch.equal(a, b) for a, b in zip(*shards)) and losses[0] == losses[1]
    worst = max(float((a - b).abs().max() / b.abs().max().clamp(min=1e-30)) for a, b in zip(*shards))
    worst = max(worst, max(abs(a - b) / max(abs(b), 1e-30) for a, b in zip(*losses)))
    flags = torch.tensor([0.0 if bit else 1.0, worst], dtype=torch.float64, device=device)
    comm.all_reduce(flags, op='max')                 # every rank must see the same verdict
    not_bit, worst = bool(flags[0].item()), float(flags[1].item())
    if not not_bit:
        verdict = 'bit-equal'
    elif worst <= 1e-6:
        verdict = 'equal within 1e-6 (reduction order)'
    else:
        verdict = 'MISMATCH'
    return {'path': 'native (hsk_shard_step on RCCL)' if verdict != 'MISMATCH' else 'none', 'verdict': verdict,
            'max_rel_diff': worst, 'steps_compared': n_steps}


# ------------------------------------------------------------------------------------------------
# one training workload
# ------------------------------------------------------------------------------------------------
def run_training(workload, device, steps, warmup, comm=None, prefetch=True, lazy_users='auto', all_stages=False,
                 pure_gather=True, strict=False):
    """-> dict(value, ms_per_step, fwd_us, fwd_launches, loss, B, N, D, data, csr, timing).  Timed exactly as the
    contract says: W warm-up steps, barrier + synchronize, K steps (+ the flush of lazily updated rows), barrier +
    synchronize; MAX over ranks."""
    from hassaku_amd.data import synthetic
    shape, D, N, B = WORKLOADS[workload]
    world = 1 if comm is None else comm.world
    data = synthetic.generate_named(shape, seed=0)          # same seed on every rank: identical data everywhere
    nnz = data.train.shape[0]
    gen = torch.Generator(device=device)
    gen.manual_seed(64)
    order = torch.randperm(nnz, device=device, generator=gen)
    if comm is not None:
        comm.broadcast(order, src=0)                              # one epoch order for the whole job
    self_check = None
    if comm is None:
        st, csr = build_state(data, D, B, N, device, overlap=prefetch, lazy_users=lazy_users)
    else:
        native = 'auto'
        if world > 1:
            # the natively issued step has to EARN being timed on this job's ranks: three steps against the phased path
            self_check = native_step_self_check(data, D, B, N, device, comm, order)
            if self_check['verdict'] == 'MISMATCH':
                if comm.rank == 0:
                    print(json.dumps({'error': 'native sharded step != phased sharded step', 'self_check': self_check}), flush=True)
                raise SystemExit(3)
            native = self_check['path'].startswith('native')
        st, csr = build_sharded_state(data, D, B, N, device, comm, native=native)
    G = B * world
    n_batches = nnz // G

    def run(n, first):
        if comm is None:
            # the epoch's inner loop, issued from C in runs of consecutive batches (hsk_bprmf_train_steps: each step
            # hints the next one to the prefetch); a run ends where the epoch order wraps around, and its last step is
            # told the batch that follows the run -- the loader always knows it -- so every step of the timed region,
            # the first and the last included, is the steady-state step
            s = 0
            while s < n:
                k0 = (first + s) % n_batches
                m = min(n - s, n_batches - k0, 256)
                nk = (k0 + m) % n_batches   # (two batches: large batches are prepared over the two steps in front of their own)
                st.hint_after_run(order, nk * B, B, N, n_batches=2 if nk + 2 <= n_batches else 1)
                st.steps_sampled(order, k0 * B, m, B, N)
                s += m
            return
        for s in range(n):
            start = ((first + s) % n_batches) * G                # global batch = world * B positives (weak scaling)
            nxt = ((first + s + 1) % n_batches) * G if s + 1 < n else None
            st.step_sampled(order, start, next_start=nxt)

    def fence():
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    # (a short --warmup leaves the clocks, the caches and the prefetch pipeline cold: at least 32 untimed steps are run)
    warmup_run = max(warmup, 32) if (comm is None and not strict) else warmup
    if comm is None and strict:
        run(warmup_run, 0)      # the driver's protocol to the letter: W warm-up steps, nothing else in front of the fence
    elif comm is None:
        # ... and the lazily updated tables are swept two steps before the fence (a flush right at the fence would drop
        # the batch the last warm-up step prepared): the timed region starts with (almost) every row current and ends
        # with every row current -- it pays for its own steps' dense-AdamW work, not for the warm-up's backlog
        run(warmup_run - 2, 0)
        st.flush()
        run(2, warmup_run - 2)
    else:
        run(warmup_run, 0)
    warmup = warmup_run
    fence()
    st.check_status('warm-up')
    # Small batches (the reference's usual 128..512) run as replayed HIP graphs, which cannot carry event records between
    # their kernels: the timed region then runs unobserved and the roofline kernel is event-timed right after it, on
    # 64 further (eager) steps of the same stream.  Large batches: events inside the timed region -- every stage on
    # every step when asked; otherwise only the roofline kernel, on every 8th step (every step for short runs), so that
    # the event records (each costs a few us of launch gap) do not distort the step time being measured.
    replayed = comm is None and (B < 2048 or os.environ.get('HSK_BENCH_REPLAYED') == '1') and not all_stages and prefetch
    names = ('prep', 'scan', 'scatter', 'fwd', 'item', 'user', 'finish') if comm is None else ('fwd', 'item', 'user')
    if not replayed:
        # an event-timed launch costs the step ~10 us (measured: 210 / 206 / 202 us per step with every 1st / 2nd / 4th
        # step timed over 20 steps): every 4th step of a short run, every 8th of a long one
        every = int(os.environ.get('HSK_BENCH_EVERY', 0)) or (1 if all_stages else max(1, steps // 3) if steps <= 64 else 8)
        st.enable_timing(names if all_stages else ('fwd',), every=every)
    ev_f0, ev_f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    run(steps, warmup)
    ev_f0.record()
    st.flush()   # lazily updated rows are brought up to date INSIDE the timed region: no work is skipped
    ev_f1.record()
    fence()
    elapsed = time.perf_counter() - t0
    flush_us = ev_f0.elapsed_time(ev_f1) * 1e3
    st.disable_timing()
    if comm is not None:                                          # the slowest rank defines the step time
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        comm.all_reduce(t, op='max')
        elapsed = float(t.item())
    n_replays = st.graph_replays() if comm is None else 0
    if replayed:
        st.enable_timing(('fwd',), every=1)
        run(64, warmup + steps)
        fence()
        st.disable_timing()
    timing = st.collect_timing()
    pure_us = None
    if comm is None and not replayed and st.st.lazy_users and pure_gather:
        # the forward as a pure gather: the pending zero-gradient AdamW steps of the batch's user rows replayed by a
        # stand-alone launch instead of inside the forward's registers (slower step, same results)
        st.st.catchup_apart = 1
        st.enable_timing(('fwd',), every=1)
        run(32, warmup + steps)
        fence()
        st.disable_timing()
        st.st.catchup_apart = 0
        ms, cnt = st.collect_timing().get('fwd', (0.0, 0))
        pure_us = ms * 1e3 / cnt if cnt else None
    st.check_status('timed region')
    loss = st.last_loss()
    assert np.isfinite(loss), loss
    fwd_ms, fwd_n = timing.get('fwd', (float('nan'), 0))
    out = dict(value=steps * B * N * world / elapsed, ms_per_step=elapsed * 1e3 / steps,
               fwd_us=(fwd_ms * 1e3 / fwd_n) if fwd_n else None, fwd_launches=int(fwd_n), loss=loss, B=B, N=N, D=D,
               data=data, csr=csr, nnz=nnz, steps=steps, warmup=warmup, graph_replays=n_replays, pure_us=pure_us,
               flush_us=flush_us, flush_cadence=st.flush_cadence(B) if comm is None else st.flush_cadence(),
               lazy_users=bool(st.st.lazy_users) if comm is None else True,
               parts=(st.batch_columns(B, N + 1) - N) if comm is None else 1, world=world, sharded=comm is not None,
               pipelined_steps=st.pipelined_steps() if comm is None else 0, self_check=self_check,
               step_issued_by=None if comm is None else st.backend() if st.issued_natively else 'python, phase by phase ('
               + st.backend() + ')')
    if all_stages:
        out['stage_us_per_step'] = {k: v[0] * 1e3 / max(v[1], 1) for k, v in timing.items()}
    del st
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------
# BASELINE configs[4]: synthetic 100 M users x 10 M items, dim = 1024, neg_train = 200, 8 ranks, both tables sharded
# ------------------------------------------------------------------------------------------------
CFG5 = dict(U=100_000_000, I=10_000_000, D=1024, N=200, B=8192, world=8, share_rank=3)


def run_cfg5(device, steps, warmup, comm=None):
    """comm with 8 ranks: the job itself -- every rank generates the interactions on its device (csrc/hsk_synth.hip: a
    pure function of the seed, no broadcast, no CSV), initialises ITS shards directly (no full table exists anywhere)
    and steps the sharded path.  comm None: ONE RANK'S SHARE on this GPU -- rank 3 of 8 with its real state (12.5 M x
    1024 user rows, items [3.75 M, 5 M), p/m/v = 169 GB, the 2e9 global interactions) and its real kernels; the
    collectives are local stand-ins (dist.LoopbackComm: the peers' user rows are stand-in rows, their score / gradient
    contributions are absent), so RCCL time is not in the figure."""
    from hassaku_amd.data.synthetic import DeviceInteractions
    from hassaku_amd.dist import LoopbackComm, ShardedBprMf, init_shard_tables
    c5 = CFG5
    U, I, D, N, B = c5['U'], c5['I'], c5['D'], c5['N'], c5['B']
    share = comm is None
    c = LoopbackComm(c5['world'], c5['share_rank']) if share else comm
    W, r = c.world, c.rank
    t_build = time.perf_counter()
    # every rank must get through the build (190 GB of allocations) or none may enter the collectives of the steps
    err = None
    try:
        need = 12.0 * D * (U / W + I / W) + 8.0 * (U + 1) + 8.5 * 20.0 * U + 4e9
        free, _ = torch.cuda.mem_get_info(device)
        if free < need:
            raise RuntimeError(f'cfg5 needs ~{need / 1e9:.0f} GB of HBM per rank, {free / 1e9:.0f} GB free')
        data = DeviceInteractions(U, I, device, seed=0)
        tabs = init_shard_tables(r, W, U, I, D, device, seed=64)
    except Exception as e:   # noqa: BLE001
        err = e
    ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=device)
    if not share:
        ok = ok.float()
        c.all_reduce(ok)
    if int(ok.item()) != (1 if share else W):
        raise RuntimeError(f'cfg5 build failed on rank {r}: {err}' if err is not None else 'cfg5 build failed on another rank')
    lazy_items = {'0': False, '1': True}.get(os.environ.get('HSK_CFG5_LAZY_ITEMS', ''), 'auto')   # (experiments)

    def all_ok(flag, what):
        """a failure on ONE rank must stop EVERY rank before the next collective (a lone rank that gives the leg up while its
        peers sit in an all_gather hangs the whole line)"""
        if share:
            if not flag:
                raise RuntimeError(what)
            return
        t = torch.tensor([1.0 if flag else 0.0], device=device)
        c.all_reduce(t)
        if int(t.item()) != W:
            raise RuntimeError(what + (' (this rank)' if not flag else ' (another rank)'))

    # the moments are allocated here (2 x the tables): past the estimate above this is where a rank could still run dry --
    # before the constructor's first collective, so every rank learns of it here
    err = None
    try:
        probe = torch.empty(int(8.0 * D * (U / W + I / W) + 2e9), dtype=torch.uint8, device=device)
        del probe
    except Exception as e:   # noqa: BLE001
        err = e
    all_ok(err is None, f'cfg5: not enough HBM for the AdamW moments: {err}')
    st = ShardedBprMf(c, tabs['user_emb'], tabs['item_emb'], tabs['item_bias'], None, None, lr=LR, wd=WD, batch=B,
                      n_neg=N, seed=64, inputs_are_shards=True, n_users=U, n_items=I, lazy_items=lazy_items,
                      **data.device_arrays())
    del tabs
    if share:
        st.rows_all.normal_(std=0.1 / D)          # the absent peers' user rows
    G = W * B
    order = data.random_order((warmup + steps + 1) * G, seed=64)
    if not share:
        c.broadcast(order, src=0)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build

    def run(n, first):
        for s in range(n):
            st.step_sampled(order, (first + s) * G, next_start=(first + s + 1) * G)

    def fence():
        c.barrier()
        torch.cuda.synchronize()

    run(warmup, 0)
    fence()
    st.check_status('cfg5 warm-up')
    st.enable_timing(('fwd', 'item', 'user'), every=1)
    fence()
    t0 = time.perf_counter()
    run(steps, warmup)
    fence()
    elapsed = time.perf_counter() - t0
    st.disable_timing()
    timing = st.collect_timing()
    if not share:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        c.all_reduce(t, op='max')
        elapsed = float(t.item())
    # the sweep of the lazily updated shards, timed on its own: it comes due every `cadence` steps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    st.flush()
    e1.record()
    torch.cuda.synchronize()
    flush_ms = e0.elapsed_time(e1)
    st.check_status('cfg5 timed region')
    loss = st.last_loss()
    offs, _, _ = st.last_batch()
    kept = int(offs[G].item())
    fu, fi = st.flush_cadence()
    stage = {k: v[0] * 1e3 / max(v[1], 1) for k, v in timing.items()}
    # k_shard_fwd: one user row per positive of the global batch (from the exchange buffer) + one item row per kept entry
    by = 4 * D * (G + kept) + 8 * kept + 12 * G
    fwd_us = stage.get('fwd')
    roof = None
    if fwd_us:
        ach = by / (fwd_us * 1e-6) / 1e9
        roof = {'bound': 'hbm', 'kernel': 'k_shard_fwd (per positive of the global batch: gather of the OWNED negatives\' item rows + scores + BPR + partial user-row gradient)',
                'achieved': ach, 'unit': 'GB/s', 'peak': HBM_PEAK_GBS, 'frac': ach / HBM_PEAK_GBS,
                'frac_of_measured_hbm_gather_5750': ach / HBM_GATHER_GBS, 'traffic': None, 'avg_us': fwd_us,
                'launches': int(timing['fwd'][1]), 'algorithmic_bytes_per_launch': by,
                'item_shard_GB': 4.0 * D * st.item_emb.shape[0] / 1e9, 'kept_entries_per_step': kept}
    # the step's DOMINANT kernel at this shape is not the gather but the item pass: 73 % of the shard's rows have an entry
    # in every global batch, so AdamW sweeps the whole shard (p, m, v read and written) and gathers a user row per entry
    item_us = stage.get('item')
    item_roof = None
    if item_us:
        I_loc = st.item_emb.shape[0]
        dense = not bool(st.sh.base.lazy_items)
        rows = I_loc if dense else I_loc * (1.0 - float(np.exp(-kept / I_loc)))   # rows with an entry in the batch
        by_i = 24 * D * rows + 4 * D * kept + 12 * kept
        ach_i = by_i / (item_us * 1e-6) / 1e9
        item_roof = {'bound': 'hbm', 'kernel': ('k_item_update_sliced' if dense else 'k_item_update_rows') + ' (item-major gradient reduction over the kept entries + AdamW on '
                     + ('every row of the shard' if dense else 'the rows with entries (lazy, exact; their catch-up before the forward is a launch of its own)') + ')',
                     'achieved': ach_i, 'unit': 'GB/s', 'peak': HBM_PEAK_GBS, 'frac': ach_i / HBM_PEAK_GBS,
                     'frac_of_measured_copy_6290': ach_i / 6290.0, 'avg_us': item_us, 'algorithmic_bytes_per_launch': by_i,
                     'share_of_step': item_us * 1e-3 / (elapsed * 1e3 / steps)}
    ms = elapsed * 1e3 / steps
    amort = (flush_ms / fu) if fu < (1 << 29) else 0.0
    out = {'workload': f'configs[4]: synthetic U={U}, I={I} ({data.nnz} interactions generated on device), mf + bpr + adamw, '
                       f'embedding_dim={D}, neg_train={N}, batch={B} per rank x {W} ranks'
                       + (f'; THIS IS RANK {r}\'S SHARE ALONE on one GPU (collectives stubbed, peers absent)' if share else ''),
           'n_gpus_of_the_job': W, 'ranks_running': 1 if share else W,
           'ms_per_step': ms, 'ms_per_step_with_amortised_sweep': ms + amort, 'steps': steps, 'warmup': warmup,
           'triplets_per_step_global': G * N, 'loss_last_step' + ('_local_share' if share else ''): loss,
           'stage_us_per_step': stage, 'roofline': roof, 'roofline_item_pass': item_roof,
           'lazy_sweep': {'cadence_steps_users': fu if fu < (1 << 29) else None,
                          'cadence_steps_items': fi if fi < (1 << 29) else None, 'sweep_ms': flush_ms,
                          'amortised_ms_per_step': amort},
           'per_rank_GB': {'tables_and_moments': 12.0 * D * (st.user_emb.shape[0] + st.item_emb.shape[0]) / 1e9,
                           'interactions': (8.0 * (U + 1) + 8.0 * data.nnz) / 1e9},
           'user_slots_per_owner': st.capacity, 'entry_capacity': st.entry_cap, 'build_seconds': t_build}
    if share:
        out['value_rank_share'] = B * N / (ms + amort) * 1e3
        out['unit'] = 'triplets/s handled by this rank (the job: x8 if the ranks overlap perfectly and RCCL hides)'
    else:
        out['value'] = G * N / (ms + amort) * 1e3
        out['unit'] = 'triplets/s'
    del st, data, order
    torch.cuda.empty_cache()
    return out


def workload_name(workload, r):
    d = r['data']
    return (f'{workload}-shaped synthetic, mf + bpr + adamw, embedding_dim={r["D"]}, neg_train={r["N"]}, '
            f'batch={r["B"]}, U={d.n_users}, I={d.n_items}, nnz_train={r["nnz"]}')


def roofline_of(workload, r):
    """Roofline object of the gather+BPR kernel for one workload.  `bound` names what the gathered tables sit in."""
    if not r['fwd_us']:
        return None
    W = r.get('world', 1)
    by = fwd_read_bytes(r['B'], r['N'], r['D'])
    if r.get('sharded'):
        # k_shard_fwd of ONE rank: a user row per positive of the GLOBAL batch (from the exchange buffer) + an item row per
        # kept entry -- the rank owns 1/W of the catalogue, so it keeps B (1 + N) of the W B (1 + N) entries on average
        by += 4 * r['D'] * r['B'] * (W - 1) + 12 * r['B'] * (W - 1)
    achieved = by / (r['fwd_us'] * 1e-6) / 1e9
    d = r['data']
    table_mb = 4.0 * r['D'] * d.n_items / 1e6
    cached = table_mb < 200.0          # 256 MiB Infinity Cache
    parts = r.get('parts', 1)
    traffic, src = pmc_traffic('k_fwd_part' if parts > 1 else 'k_fwd_ugrad', workload)
    if parts > 1:
        # item-partitioned forward (csrc/hsk_fwd_part.h): an XCD gathers from its 1/P of the item table only, most of it
        # resident in its own 4 MB L2 -- the bound is the L2s' gather rate
        bound, peak = 'l2', L2_GATHER_GBS
        kernel = ('k_fwd_part, P = %d (gather + scores + BPR + partial user-row grads; the next batch\'s lazy user rows brought up to '
                  'date by workgroups of the same launch; the launch also carries the sampler of batch t+2 and two sort phases of '
                  'batches t / t+1: about 2.5 us of avg_us, MEASUREMENTS.md R4.2b)') % parts
        peak_source = ('MI355X_MICROARCH.md, "Indexed rows": 16.8-18.8 TB/s chip-wide for rows served by the XCDs\' L2s (mid-point); '
                       'each XCD gathers from %.1f MB of the %.1f MB item table' % (table_mb / parts, table_mb))
    elif r.get('sharded') and table_mb / W <= 4.0:
        # a rank's range shard of the item table fits every XCD's 4 MB L2
        bound, peak = 'l2', L2_GATHER_GBS
        kernel = 'k_shard_fwd'
        peak_source = ('MI355X_MICROARCH.md, "Indexed rows": 16.8-18.8 TB/s chip-wide for rows served by the XCDs\' L2s (mid-point); '
                       'this rank gathers from its %.1f MB shard of the %.1f MB item table' % (table_mb / W, table_mb))
    elif cached:
        bound, peak = 'infinity-cache', ICACHE_GATHER_GBS
        kernel = 'k_fwd_ugrad (gather + scores + BPR + user-row grad)'
        peak_source = ('MI355X_MICROARCH.md: 8.6 TB/s measured for uniformly random row gathers from an Infinity-Cache-'
                       'resident table (the %.1f MB item table is cache-resident; FETCH_SIZE counts cache hits, so true '
                       'HBM bytes of this kernel are not observable)' % table_mb)
    else:
        bound, peak = 'hbm', HBM_PEAK_GBS
        kernel = 'k_fwd_ugrad (gather + scores + BPR + user-row grad)'
        peak_source = 'MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (guide measures 6.29 TB/s streaming, 5.7-5.8 TB/s for random 2.3 KB rows)'
    out = {'bound': bound, 'kernel': kernel, 'achieved': achieved, 'unit': 'GB/s', 'peak': peak, 'frac': achieved / peak,
           'peak_source': peak_source,
           'frac_of_hbm_spec_8000': achieved / HBM_PEAK_GBS,
           'traffic': traffic, 'traffic_source': src,
           'avg_us': r['fwd_us'], 'launches': r['fwd_launches'], 'algorithmic_bytes_per_launch': by,
           'item_table_MB': table_mb}
    if parts > 1:
        out['frac_of_infinity_cache_gather_8600'] = achieved / ICACHE_GATHER_GBS
        fetch = pmc_fetch_bytes('k_fwd_part', workload)
        if fetch and fetch < by:
            # A MODEL, not a measurement of this run: the XCD's partition (table / P) is larger than its 4 MB L2, so part of
            # the gathered bytes misses the L2 and crosses the fabric from the Infinity Cache.  Bytes the counters saw on
            # the fabric side priced at the guide's Infinity-Cache gather rate, the rest at its L2 gather rate:
            t_mixed = fetch / (ICACHE_GATHER_GBS * 1e9) + (by - fetch) / (L2_GATHER_GBS * 1e9)
            out['mixed_l2_fabric_model'] = {
                'l2_miss_share': fetch / by, 'floor_us': t_mixed * 1e6, 'frac': t_mixed / (r['fwd_us'] * 1e-6),
                'l2_hit_rate_counters': pmc_fetch_bytes('k_fwd_part', workload, 'l2_hit_rate'),   # all requests, writes included
                'pure_gather_frac': (t_mixed / (r['pure_us'] * 1e-6)) if r.get('pure_us') else None,
                'note': 'arithmetic from the committed profile\'s FETCH_SIZE (x2) and the guide\'s two gather rates; `frac` above, '
                        'against the all-hits L2 rate, stays the quoted roofline fraction'}
    if not cached:
        out['frac_of_measured_hbm_gather_5750'] = achieved / HBM_GATHER_GBS
    if r.get('pure_us'):
        # avg_us above includes the replay of lazily updated user rows' pending AdamW steps (VALU work: in the
        # wave's registers, or -- partitioned forward -- by workgroups of the same launch for the NEXT batch's rows;
        # ~9-12 us at the ml10m shape); with that replay in a launch of its own the kernel is the gather alone:
        pa = by / (r['pure_us'] * 1e-6) / 1e9
        out['pure_gather'] = {'avg_us': r['pure_us'], 'achieved': pa, 'frac': pa / out['peak'],
                              'frac_of_hbm_spec_8000': pa / HBM_PEAK_GBS,
                              'note': 'st.catchup_apart = 1, 32 eager steps after the timed region'}
    # the metric's own roofline: whole step against algorithmic gather bytes at the HBM spec peak
    out['step_frac_of_hbm_roofline'] = r['value'] / W / (HBM_PEAK_GBS * 1e9 / (fwd_read_bytes(r['B'], r['N'], r['D']) / (r['B'] * r['N'])))
    return out


# ------------------------------------------------------------------------------------------------
# evaluation legs
# ------------------------------------------------------------------------------------------------
class _EvalData:
    """Synthetic exclusion / ground-truth CSRs of one evaluation shape (what a FullEvalDataset exposes)."""

    def __init__(self, U, I, npos, device):
        from hassaku_amd.data.csr import UserItemCsr
        rng = np.random.default_rng(0)
        users = np.repeat(np.arange(U), npos)
        self.exclude_csr = UserItemCsr.from_pairs(users, rng.integers(0, I, size=len(users)), U, I)
        self.label_csr = UserItemCsr.from_pairs(np.repeat(np.arange(U), 10), rng.integers(0, I, size=10 * U), U, I)
        self.n_users, self.n_items = U, I
        lp, li = self.label_csr.to_device(device)
        ep, ei = self.exclude_csr.to_device(device)
        self._arr = {'label_indptr': lp, 'label_indices': li, 'excl_indptr': ep, 'excl_indices': ei}

    def device_arrays(self, device):
        return self._arr


def eval_problem(shape, device):
    """(user_emb, item_emb, item_bias, dataset) of one evaluation leg -- also what tests/test_hip_parity.py pins against
    the oracle (test_bench_eval_pass_vs_oracle): the pass the bench times is the pass the test checks."""
    U, I, D, npos = EVAL_SHAPES[shape]
    torch.manual_seed(0)
    user_emb = torch.randn(U, D, device=device) * 0.05
    item_emb = torch.randn(I, D, device=device) * 0.05
    item_bias = torch.randn(I, device=device) * 0.1
    return user_emb, item_emb, item_bias, _EvalData(U, I, npos, device)


def eval_arithmetic(tf, world):
    form = os.environ.get('HSK_EVAL_X3', '2')
    if form == '0':
        return {'arithmetic': 'exact-fp32 MFMA (HSK_EVAL_X3=0)'}
    if form == '1':
        return {'arithmetic': 'fp32-accurate scores from three bf16 pieces per operand, six bf16 MFMAs per block (HSK_EVAL_X3=1)',
                'frac_of_bf16_six_product_peak_417': tf / (MFMA_BF16_TFLOPS / 6.0 * world)}
    return {'arithmetic': 'fp32-accurate scores from two scaled fp16 pieces per operand, three fp16 MFMAs per block '
                          '(default; HSK_EVAL_X3=1: three bf16 pieces / six MFMAs, 0: exact-fp32 MFMA)',
            'frac_of_f16_three_product_peak_833': tf / (MFMA_BF16_TFLOPS / 3.0 * world)}


def run_eval(shape, device, comm=None, chunk=None, repeat=3):
    """Full-catalogue evaluation (scores U x I^T, exclusion mask, top-100, precision/recall/ndcg at 100/50/10/5).
    comm None: one GPU.  Otherwise ITEM-SHARDED over the ranks on physically sliced tables (dist.evaluate_item_sharded)."""
    from hassaku_amd import hip_ops as ops
    from hassaku_amd.eval.eval import FullEvaluator
    U, I, D, npos = EVAL_SHAPES[shape]
    if chunk is None:
        # eval_batch_size (the reference's conf key; the caller's choice).  Wide catalogue, top-k inside the GEMM: 16 384
        # users.  Narrow, materialised scores: 1.5 GB of them per chunk (35 072 users at the ml10m width: the score GEMM's
        # 256 x 256 workgroups then fill their last round of 256 CUs better and the item table is cut into its pieces twice
        # per pass instead of five times -- 13.4 -> 14.0 M users/s against chunks of 16 384)
        rule = 16384 if I >= ops.FUSED_TOPK_MIN_ITEMS else max(256, min(U + 255, int(1.5e9 / (4 * I))) // 256 * 256)
        chunk = int(os.environ.get('HSK_BENCH_EVAL_CHUNK', rule))
    user_emb, item_emb, item_bias, ds = eval_problem(shape, device)
    ev = FullEvaluator(aggr_by_group=True, n_groups=0, user_to_user_group=None)
    ks = sorted(ev.K_VALUES, reverse=True)
    world = 1 if comm is None else comm.world
    if comm is None:
        arr = ds.device_arrays(device)

        def one_pass():
            acc = torch.zeros((len(ks), 3), dtype=torch.float64, device=device)
            for lo in range(0, U, chunk):
                u = torch.arange(lo, min(lo + chunk, U), device=device)
                _, ids, _ = ops.mf_eval_topk(user_emb, item_emb, item_bias, None, None, u, ks[0], arr['excl_indptr'],
                                             arr['excl_indices'])
                acc += ops.rank_metrics(ids, u, arr['label_indptr'], arr['label_indices'], ks).double().sum(0)
            return {f'ndcg@{k}': float(acc[t, 2] / U) for t, k in enumerate(ks)}
    else:
        from hassaku_amd.dist import TableShards, evaluate_item_sharded
        shards = TableShards.cut(comm, user_emb, item_emb, item_bias)
        del user_emb, item_emb, item_bias

        chunk = min(chunk, U)

        def one_pass():
            return evaluate_item_sharded(comm, shards, ds, ev, chunk=chunk)

    check = one_pass()                # warm-up
    torch.cuda.synchronize()
    if comm is not None:
        comm.barrier()
    t0 = time.perf_counter()
    for _ in range(repeat):
        one_pass()
    torch.cuda.synchronize()
    if comm is not None:
        comm.barrier()
    dt = (time.perf_counter() - t0) / repeat
    if comm is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        comm.all_reduce(t, op='max')
        dt = float(t.item())
    tf = 2.0 * U * I * D / dt / 1e12
    return {'workload': f'{shape}-shaped full evaluation: U={U}, I={I}, D={D}, top-100 + metrics at 100/50/10/5, '
                        f'{npos} excluded positives per user, chunks of {chunk} users',
            'n_gpus': world, 'sharding': 'items (range-sharded tables, candidate all_to_all)' if comm is not None else 'none',
            'users_per_s': U / dt, 'seconds_per_full_eval': dt, 'tflops_fp32': tf,
            'frac_of_fp32_mfma_peak_157': tf / (MFMA_FP32_TFLOPS * world),
            # the score GEMMs run on the 16-bit matrix cores at fp32-GEMM accuracy: form 2 (default) two fp16 pieces per
            # operand, three MFMAs per product block (csrc/hsk_gemm_wide_h2.h) -- 2500 / 3 TFLOP/s fp32-equivalent; form 1
            # three bf16 pieces, six MFMAs (2500 / 6); form 0 the exact-fp32 MFMA (157)
            **eval_arithmetic(tf, world),
            'ndcg@10_check': check['ndcg@10']}


# ------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------
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


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(headline, budget_s):
    """oracle/cpu_trainer.py (kind "port") on the host cores, BEFORE the process touches the GPU (the 4-worker leg
    forks loader processes).  Headline: a bounded sample of the ml10m workload.  Legs: SURVEY 8(d)'s protocol on
    configs[0] (ml100k shape, D=64, N=1, B=128): 20 warm-up + 200 timed steps with 0 and with 4 loader workers."""
    from hassaku_amd.data import synthetic
    from hassaku_amd.data.csr import UserItemCsr
    from oracle.cpu_trainer import CpuTrainer
    cores = usable_cores()

    def trainer(workload):
        shape, D, N, B = WORKLOADS[workload]
        data = synthetic.generate_named(shape, seed=0)
        csr = UserItemCsr.from_pairs(data.train[:, 0], data.train[:, 1], data.n_users, data.n_items)
        return CpuTrainer(data.n_users, data.n_items, D, LR, WD, csr.indptr, csr.indices, data.train[:, 0],
                          data.train[:, 1], N, B, threads=cores), D, N, B

    legs = []
    tr, D, N, B = trainer('ml100k')
    for workers in (0, 4):
        steps, secs = tr.time_loader_steps(workers=workers, warmup=20, steps=200)
        legs.append({'workload': f'configs[0]: ml100k shape, D={D}, N={N}, B={B}', 'train_n_workers': workers,
                     'warmup': 20, 'steps': steps, 'seconds': secs, 'value': steps * B * N / secs, 'unit': 'triplets/s'})
    tr, D, N, B = trainer(headline)
    steps, secs = tr.time_steps(budget_s=budget_s)
    return {'value': steps * B * N / secs, 'unit': 'triplets/s', 'cores': cores, 'cpu_model': cpu_model(),
            'kind': 'port',
            'sample': f'{steps} steps of B={B} x N={N} (D={D}, {headline} shape) in {secs:.1f}s, torch CPU ops on '
                      f'{cores} threads + numpy rejection sampler, 0 loader workers',
            'legs': legs}


def child_bench(device, *args, env_extra=None):
    """`python bench.py --cpu-budget 0 <args>` as a child process on the same GPU -> its JSON line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-budget', '0'] + list(args)
    env = dict(os.environ, MASTER_PORT=os.environ.get('HSK_BENCH_CHILD_PORT', '29541'), RANK='0', WORLD_SIZE='1',
               LOCAL_RANK=str(device.index or 0), **(env_extra or {}))
    torch.cuda.empty_cache()
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    except subprocess.TimeoutExpired as e:
        tail = (e.stderr or b'')[-300:] if isinstance(e.stderr, (bytes, bytearray)) else str(e.stderr or '')[-300:]
        raise RuntimeError(f'child bench killed at its 900 s limit; stderr tail: {tail!r}')
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    if p.returncode != 0 or not lines:
        raise RuntimeError(f'child bench failed (rc {p.returncode}): {p.stderr[-300:]}')
    return json.loads(lines[-1])


def run_ieee_build_child(device, steps, warmup):
    """What the default build's Adam arithmetic buys: the same step with libhassaku_hip_ieee.so (-DHSK_ADAM_IEEE=1: IEEE sqrt
    and divisions in the optimiser, torch's own arithmetic bit for bit) in a child process, on configs[2] under the same
    protocol and on the HBM-resident point.  The default build uses v_sqrt_f32 / v_rcp_f32 (1 ulp); both builds meet the
    same parity bounds (tests/test_hip_parity.py::test_default_build_vs_ieee_build_on_golden_steps)."""
    from hassaku_amd import _lib
    ieee = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libhassaku_hip_ieee.so')
    if not os.path.isfile(ieee):
        raise RuntimeError('libhassaku_hip_ieee.so not built')
    env = {'HSK_LIB_PATH': ieee}
    a = child_bench(device, '--only', '--no-pure-gather', '--steps', str(steps), '--warmup', str(warmup), env_extra=env)
    b = child_bench(device, '--only', '--no-pure-gather', '--workload', 'hbm', '--steps', '64', '--warmup', '16', env_extra=env)
    return {'build': '-DHSK_ADAM_IEEE=1 (IEEE sqrt / divisions in every optimiser update and replay)',
            'ml10m': {'ms_per_step': a['ms_per_step'], 'steps': steps, 'warmup': warmup, 'fwd_us': a['roofline']['avg_us'],
                      'flush_us_in_timed_region': a['flush_us_in_timed_region']},
            'hbm': {'ms_per_step': b['ms_per_step'], 'steps': 64, 'warmup': 16, 'fwd_us': b['roofline']['avg_us']}}


def run_cfg5_share_child(device):
    """one rank's share of configs[4] in a process of its own (190 GB of HBM, and hardware queues of its own: see
    run_sharded_1rank)"""
    return child_bench(device, '--workload', 'cfg5', '--steps', '12', '--warmup', '4')['workloads']['cfg5_shard']


def run_sharded_1rank(workload, device):
    """In a child process of its own: the leg's three streams (step, preparation, exchange) then get hardware queues of
    their own -- late in THIS process, after the other legs' streams, they end up sharing queues (ROCm maps streams onto
    a handful of hardware queues in creation order) and the overlap the step is built on is lost: 530 us per step here
    against 290 in a fresh process, same binary, same box."""
    x = child_bench(device, '--sharded', '--only', '--steps', '200', '--warmup', '20', '--workload', workload, '--with-eval')
    return {'workload': x['config']['workload'] + '; ShardedBprMf on a 1-rank RCCL group in a process of its own '
                                                  '(hsk_shard_step: the whole step from one C call, collectives on the '
                                                  'library\'s own communicator)',
            'value': x['value'], 'unit': 'triplets/s', 'steps': x['steps'], 'warmup': x['warmup'],
            'ms_per_step': x['ms_per_step'], 'fwd_us': x['roofline']['avg_us'],
            'loss_last_step': x['config']['loss_last_step'],
            'eval_lfm2b_item_sharded_users_per_s': (x.get('eval') or {}).get('lfm2b', {}).get('users_per_s')}


XGMI_LINK_GBS_PER_DIRECTION = 76.5   # guide: 7 links x ~153 GB/s per GPU, bidirectional -> per link and direction


def predicted_scaling(sharded_us, B, N, D, U, item_us=90.0, user_update_us=55.0, small_us=15.0):
    """PREDICTION, not a measurement (no multi-GPU node has run this code): the weak-scaling step from the measured
    one-rank sharded step and the guide's link rate.  The mesh is point to point: a rank sends its C user rows to each of
    its W - 1 peers over that peer's own link, so a row exchange takes 4 D C / link rate whatever W is (C = user slots per
    owner ~ B + 6 sigma + 8).  As built: the all_gather opens the step, exposed in full; the reduce_scatter and, behind it,
    the owners' user update run beside the item pass and are exposed by what they outlast it; two small reductions cost
    their latency.  `pipelined`: the design of MEASUREMENTS.md R4.6 -- the rows of batch t+1 that batch t does not train
    on are gathered during step t, only the shared ones (a fraction 1 - exp(-G / U) of them) after the owners' update --
    not built.  Both exchanges of a step cross every link in the same direction: 2 x row_exchange_us is a floor."""
    import math
    out = {'note': 'PREDICTED from the measured 1-rank sharded step + %.1f GB/s per xGMI link and direction; weak scaling, '
                   'B = %d per rank; nothing here was measured on more than one GPU' % (XGMI_LINK_GBS_PER_DIRECTION, B),
           'one_rank_sharded_us': sharded_us, 'item_pass_us_assumed': item_us, 'user_update_us_assumed': user_update_us,
           'small_all_reduce_us_assumed': small_us}
    for W in (2, 4, 8):
        G = W * B
        C = G / W + 6.0 * math.sqrt(G / W * (1.0 - 1.0 / W)) + 8
        t_x = 4.0 * D * C / (XGMI_LINK_GBS_PER_DIRECTION * 1e3)        # us per row exchange
        tail = max(0.0, t_x + user_update_us - item_us)                # reduce_scatter + user update beyond the item pass
        as_built = sharded_us + t_x + tail + 2 * small_us
        shared = 1.0 - math.exp(-G / float(U))                         # rows of batch t+1 that batch t updates
        piped = max(2.0 * t_x, sharded_us - 27.0 + tail + shared * t_x + 3.0 + 2 * small_us)   # catch-up + pack + gather off the path
        out[f'W{W}'] = {'row_exchange_us': t_x, 'rows_shared_with_previous_batch': shared,
                        'step_us_as_built': as_built, 'triplets_per_s_as_built': G * N / as_built * 1e6,
                        'step_us_pipelined_design': piped, 'triplets_per_s_pipelined_design': G * N / piped * 1e6}
    return out


def guarded(fn, *a):
    """An extra leg must not cost the line its headline: a failure is recorded, not raised."""
    try:
        return fn(*a)
    except Exception as e:   # noqa: BLE001
        torch.cuda.empty_cache()
        return {'error': f'{type(e).__name__}: {e}'[:500]}


# ------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='ml10m', choices=sorted(WORKLOADS) + ['cfg5'],
                    help="cfg5 = BASELINE configs[4] (100 M x 10 M, D=1024): the job itself with --gpus 8, one rank's "
                         "share of it (rank 3 of 8, collectives stubbed) with --gpus 1")
    ap.add_argument('--cpu-budget', type=float, default=12.0, help='seconds of CPU-baseline work on the headline workload (0 = skip every CPU leg)')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for --gpus > 1 ('nccl' = RCCL; 'gloo' "
                    "stages collectives through the host and lets several ranks share one GPU: functional rehearsal only)")
    ap.add_argument('--no-prefetch', action='store_true', help='do not prepare (sample + sort) the next batch on a side stream during the current step')
    ap.add_argument('--dense-users', action='store_true', help='dense AdamW sweep over the user table every step')
    ap.add_argument('--lazy-users', action='store_true', help='lazy, exact user AdamW whatever the table size')
    ap.add_argument('--time-all-stages', action='store_true', help='event-time every stage (perturbs the step time)')
    ap.add_argument('--only', action='store_true', help='headline workload only: no extra workloads, no eval legs')
    ap.add_argument('--no-pure-gather', action='store_true', help='skip the extra pure-gather timing pass (profiling: keeps the kernel averages of the timed region unmixed)')
    ap.add_argument('--eval-only', default=None, choices=sorted(EVAL_SHAPES), help='only the evaluation leg of this shape (profiling)')
    ap.add_argument('--sharded', action='store_true', help='N=1 through the multi-GPU code path (1-rank process group)')
    ap.add_argument('--with-eval', action='store_true', help='with --only: the item-sharded lfm2b evaluation leg as well')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')

    cpu = None
    if world == 1 and args.cpu_budget > 0:
        cpu = cpu_baseline(args.workload, args.cpu_budget)        # before anything initialises the GPU

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (no CPU fallback)')
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    comm = None
    if world > 1 or args.sharded:
        import torch.distributed as dist
        from hassaku_amd.dist import Comm
        if world == 1:
            os.environ.setdefault('MASTER_PORT', '29533')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(args.backend)
        comm = Comm()

    if args.workload == 'cfg5':
        if world not in (1, CFG5['world']):
            raise SystemExit('--workload cfg5 runs with --gpus 8 (the job) or --gpus 1 (one rank\'s share)')
        x = run_cfg5(device, args.steps, args.warmup, comm if world > 1 else None)
        out = {'workloads': {'cfg5' if world > 1 else 'cfg5_shard': x}}
        if rank == 0:
            print(json.dumps(out), flush=True)
        if comm is not None:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.eval_only:
        out = {'eval': {args.eval_only: run_eval(args.eval_only, device, comm)}}
        if rank == 0:
            print(json.dumps(out), flush=True)
        if comm is not None:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return
    # The two legs that run in child processes go FIRST, while this process holds nothing on the device: one rank's
    # 169 GB share of configs[4] gathers random 4 KB rows out of tables that large, and it ran 9 % slower (item pass 5.57
    # against 4.92 ms) as a child of a parent that had already been through its other legs than from a fresh process.
    children = {}
    if comm is not None and world == CFG5['world'] and not args.only and args.workload != 'cfg5':
        # (the same for the one place the whole configs[4] job can run: before this process's other legs)
        children['cfg5'] = guarded(run_cfg5, device, 12, 4, comm)
    if comm is None and not args.only:
        # BASELINE configs[4] needs 8 GPUs; what one GPU can show is one rank's full-size share of it
        children['cfg5_shard'] = guarded(run_cfg5_share_child, device)
        # the multi-GPU code path on this one GPU (a 1-rank RCCL group: every kernel, every collective call and stream
        # hand-off of the sharded step, no link traffic): what the step costs before any xGMI link is involved
        children['sharded_1rank'] = guarded(run_sharded_1rank, args.workload, device)
        children['ieee_build'] = guarded(run_ieee_build_child, device, args.steps, args.warmup)
    r = run_training(args.workload, device, args.steps, args.warmup, comm=comm, prefetch=not args.no_prefetch,
                     lazy_users=False if args.dense_users else True if args.lazy_users else 'auto',
                     all_stages=args.time_all_stages, pure_gather=not args.no_pure_gather)
    out = {
        'metric': 'BPR triplets/sec', 'value': r['value'], 'unit': 'triplets/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': r['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': workload_name(args.workload, r), 'global_batch': r['B'] * world,
                   'parallelism': 'single GPU' if comm is None else
                   f'item-sharded: {world} rank(s), item table range-sharded + user table row-sharded; all_gather of '
                   f'user rows, reduce_scatter of user-row gradients, 2 scalar-per-positive all_reduces; no table is replicated',
                   'lr': LR, 'wd': WD, 'loss_last_step': r['loss'],
                   'user_adamw': 'lazy, exact' if r['lazy_users'] else 'dense sweep',
                   'steps_issued_as_replayed_graphs': 64 * r['graph_replays'],
                   'steps_with_in_launch_preparation': r['pipelined_steps'],
                   'warmup_steps_run': r['warmup'],
                   'protocol_note': 'at least 32 untimed steps run whatever --warmup says, and the lazily updated tables are '
                                    'swept two steps before the fence (the timed region pays for its own dense-AdamW backlog, '
                                    'not the warm-up\'s); `as_given_protocol` is the same workload with exactly --warmup '
                                    'steps and nothing else in front of the fence'},
        # the closing sweep of the lazily updated tables is inside the timed region: ms_per_step carries 1/steps of it
        'flush_us_in_timed_region': r['flush_us'],
        'ms_per_step_without_closing_flush': r['ms_per_step'] - r['flush_us'] * 1e-3 / args.steps,
        'lazy_sweep_cadence_steps': [None if f >= (1 << 29) else f for f in r['flush_cadence']],
        'roofline': roofline_of(args.workload, r),
    }
    if comm is not None:
        out['roofline']['kernel'] = 'k_shard_fwd (gather of the owned negatives + scores + BPR + partial user-row grad)'
        out['config']['step_issued_by'] = r['step_issued_by']
        if r['self_check'] is not None:
            out['config']['native_vs_phased_self_check'] = r['self_check']
    if 'stage_us_per_step' in r:
        out['stage_us_per_step'] = r['stage_us_per_step']

    if args.only and args.with_eval and comm is not None:
        out['eval'] = {'lfm2b': run_eval('lfm2b', device, comm)}
    if not args.only:
        if comm is None:
            # the other BASELINE training configs + the HBM-resident point, each on the same clock as the headline
            out['workloads'] = {}
            for name, (k, w) in (('ml1m', (1920, 192)), ('ml100k', (1920, 192)), ('hbm', (64, 16))):
                if name == args.workload:
                    continue
                x = run_training(name, device, k, w)
                out['workloads'][name] = {
                    'workload': workload_name(name, x), 'value': x['value'], 'unit': 'triplets/s', 'steps': k,
                    'warmup': w, 'ms_per_step': x['ms_per_step'], 'loss_last_step': x['loss'],
                    'user_adamw': 'lazy, exact' if x['lazy_users'] else 'dense sweep',
                    'steps_issued_as_replayed_graphs': 64 * x['graph_replays'],
                    'roofline': roofline_of(name, x)}
            out['eval'] = {s: run_eval(s, device) for s in ('ml10m', 'lfm2b')}
            out['workloads']['cfg5_shard'] = children['cfg5_shard']
            out['sharded_1rank'] = children['sharded_1rank']
            if 'ms_per_step' in children['sharded_1rank']:
                out['predicted_scaling'] = predicted_scaling(children['sharded_1rank']['ms_per_step'] * 1e3, r['B'], r['N'],
                                                             r['D'], r['data'].n_users)
            out['ieee_build'] = children['ieee_build']
            # the driver's protocol to the letter: exactly --warmup steps, no sweep in front of the fence
            x = guarded(run_training, args.workload, device, args.steps, args.warmup, None, not args.no_prefetch, 'auto',
                        False, False, True)
            out['as_given_protocol'] = x if 'error' in x else {
                'ms_per_step': x['ms_per_step'], 'value': x['value'], 'unit': 'triplets/s', 'steps': args.steps,
                'warmup_steps_run': x['warmup'], 'flush_us_in_timed_region': x['flush_us'],
                'note': 'cold clocks / caches and the warm-up\'s whole lazy-AdamW backlog are inside this timed region'}
        else:
            out['eval'] = {'lfm2b': run_eval('lfm2b', device, comm)}
            if 'cfg5' in children:
                out['workloads'] = {'cfg5': children['cfg5']}
    if cpu is not None:
        out['cpu_baseline'] = cpu
    if rank == 0:
        print(json.dumps(out), flush=True)
    if comm is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
