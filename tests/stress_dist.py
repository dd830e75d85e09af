#!/usr/bin/env python3
"""Randomised check of the sharded multi-rank step against the single-GPU step (test infrastructure; not collected):
    python tests/stress_dist.py [seconds] [seed] [world]          (on a GPU box; the ranks share GPU 0)
`world` processes over gloo, the natively issued step (hsk_shard_step) on the host-staged collective table or the phased
step; random shapes (D, negatives, table sizes, per-rank batch), losses, lazy / dense item AdamW, with and without the
ahead-of-time preparation, a wrong next-batch guess now and then.  Rank 0 compares the gathered tables and the losses with
BprMfFusedState on the same global batches (same seed / order / step numbering -> the same samples) under the tolerance
rules of tests/conftest.py; in 40 % of the cases the item-sharded evaluation follows and is compared with the single-GPU
evaluation of the gathered tables.  (Last runs: 5560 cases, 2189 with the evaluation; 3 reports, all the worst parameter
element at 2.2-2.6e-4 of the table's largest against the rule's 2e-4 -- the ranks' partial sums add up in another order.)"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


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


def worker(rank, world, port, budget, seed):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from conftest import assert_adam_param_close, csr_from_pairs, max_norm_err
    from hassaku_amd import hip_ops as ops
    from hassaku_amd.data.csr import UserItemCsr
    from hassaku_amd.dist import Comm, ShardedBprMf, evaluate_item_sharded
    from hassaku_amd.eval.eval import FullEvaluator
    comm = Comm()
    rng = np.random.RandomState(seed)          # the same stream on every rank
    t_end = time.time() + budget
    n = bad = skipped = n_eval = amplified = redo = 0
    while True:
        go = torch.tensor([1 if time.time() < t_end else 0])
        dist.broadcast(go, 0)
        if not int(go.item()):
            break
        D = int(rng.choice([8, 30, 33, 64, 100, 128, 256, 402, 512]))
        N = int(rng.choice([1, 3, 12, 20, 50]))
        U, I = int(rng.randint(40, 600)), int(rng.randint(N + 30, 1500))
        B = int(rng.choice([8, 32, 48, 100, 256]))
        loss = str(rng.choice(['bpr', 'bpr', 'bce', 'sampled_softmax']))
        lazy_items = ['auto', True, False][int(rng.randint(3))]
        prefetch = bool(rng.rand() < 0.7)
        native = 'host-staged' if rng.rand() < 0.7 else False
        n_steps = int(rng.randint(3, 30))
        pairs = np.argwhere(rng.rand(U, I) < min(0.3, 25.0 / I))
        wrong_at = int(rng.randint(0, n_steps))
        do_eval = bool(rng.rand() < 0.4) and I >= 100
        val = np.argwhere(rng.rand(U, I) < min(0.2, 8.0 / I))
        P = {'user_emb': (rng.randn(U, D) * 0.05).astype(np.float32), 'item_emb': (rng.randn(I, D) * 0.05).astype(np.float32),
             'item_bias': (rng.randn(I) * 0.1).astype(np.float32)}
        G = world * B
        if len(pairs) < 2 * G:
            continue
        pairs = pairs[rng.permutation(len(pairs))]
        desc = dict(D=D, N=N, U=U, I=I, B=B, loss=loss, lazy_items=lazy_items, prefetch=prefetch, native=native, n_steps=n_steps)
        ptr, idx = csr_from_pairs(pairs, U)
        adj = float(np.log(I / N)) if loss == 'sampled_softmax' else 0.0
        order = torch.from_numpy(np.random.RandomState(1).permutation(len(pairs))).cuda()
        n_slices = len(pairs) // G
        def run_sharded(k_steps, with_eval):
            t = {k: _dev(v) for k, v in P.items()}
            sh = ShardedBprMf(comm, t['user_emb'], t['item_emb'], t['item_bias'], None, None, lr=2e-3, wd=1e-4, batch=B, n_neg=N,
                              csr_indptr=_dev(ptr), csr_indices=_dev(idx), coo_user=_dev(pairs[:, 0], torch.int32),
                              coo_item=_dev(pairs[:, 1], torch.int32), seed=77, lazy_items=lazy_items, prefetch=prefetch,
                              native=native, loss=loss, log_adjust=adj)
            losses = []
            for s in range(k_steps):
                nxt = ((s + 1) % n_slices) * G if s + 1 < k_steps else None
                if s == wrong_at:
                    nxt = 0
                sh.step_sampled(order, (s % n_slices) * G, next_start=nxt)
                losses.append(sh.last_loss())
            sh.check_status()
            full_u, _ = sh.gather_user_table()
            full_i, full_ib = sh.gather_item_table()
            full_mi, full_vi = sh.gather_item_moments()
            metrics = None
            if with_eval:
                metrics = evaluate_item_sharded(comm, sh, _EvalDs(pairs, val, U, I),
                                                FullEvaluator(aggr_by_group=False, n_groups=0, user_to_user_group=None), chunk=64)
            sh.close()
            return losses, full_u, full_i, full_ib, full_mi, full_vi, metrics

        def run_single(k_steps):
            tt = {k: _dev(v) for k, v in P.items()}
            st = ops.BprMfFusedState(tt['user_emb'], tt['item_emb'], tt['item_bias'], None, None, lr=2e-3, wd=1e-4, max_batch=G,
                                     max_cols=N + 1, seed=77, csr_indptr=_dev(ptr), csr_indices=_dev(idx),
                                     coo_user=_dev(pairs[:, 0], torch.int32), coo_item=_dev(pairs[:, 1], torch.int32),
                                     lazy_items=lazy_items, loss=loss, log_adjust=adj)
            ref_losses = []
            for s in range(k_steps):
                st.step_sampled(order, (s % n_slices) * G, G, N)
                ref_losses.append(st.last_loss())
            st.flush()
            return st, tt, ref_losses

        failed = None
        try:
            losses, full_u, full_i, full_ib, full_mi, full_vi, metrics = run_sharded(n_steps, do_eval)
        except RuntimeError as e:          # a capacity overflow is raised on every rank alike: not a parity case
            failed = str(e)[:200]
        flag = torch.tensor([1 if failed else 0])
        dist.all_reduce(flag)
        if int(flag.item()):
            skipped += 1
            if rank == 0 and skipped <= 5:
                print('skipped', desc, failed, flush=True)
            continue
        if rank == 0:
            n += 1
            st, tt, ref_losses = run_single(n_steps)
            try:
                np.testing.assert_allclose(np.array(losses), np.array(ref_losses), rtol=2e-5)
                assert_adam_param_close(full_u.cpu().numpy(), tt['user_emb'].cpu().numpy(), 'user_emb')
                assert_adam_param_close(full_i.cpu().numpy(), tt['item_emb'].cpu().numpy(), 'item_emb')
                assert_adam_param_close(full_ib.cpu().numpy(), tt['item_bias'].cpu().numpy(), 'item_bias')
                if metrics is not None:   # item-sharded evaluation == single-GPU evaluation of the gathered tables
                    lab = UserItemCsr.from_pairs(val[:, 0], val[:, 1], U, I)
                    exc = UserItemCsr.from_pairs(pairs[:, 0], pairs[:, 1], U, I)
                    lp, lix = lab.to_device('cuda')
                    ep, ei = exc.to_device('cuda')
                    uu = torch.arange(U, device='cuda')
                    ks = [100, 50, 10, 5]
                    _, ids, _ = ops.mf_eval_topk(full_u, full_i, full_ib, None, None, uu, min(100, I), ep, ei)
                    if ids.shape[1] == 100:
                        met = ops.rank_metrics(ids, uu, lp, lix, ks).double().cpu().numpy()
                        for ti, k in enumerate(ks):
                            for j, name in enumerate(('precision', 'recall', 'ndcg')):
                                assert abs(metrics[f'{name}@{k}'] - met[:, ti, j].mean()) < 1e-6, (name, k, metrics[f'{name}@{k}'])
                    n_eval += 1
            except AssertionError as e:
                # Adam divides by sqrt(v) + eps: where a gradient element cancels to <~ eps, two summation orders (the
                # ranks' entry lists against the single GPU's) move the PARAMETER apart by up to ~lr while exp_avg /
                # exp_avg_sq, linear / quadratic in g, stay together (tests/conftest.py).  Both moments of the item table
                # within 1e-5 of their largest element: reported as such, not as a failure.
                em = max_norm_err(full_mi.cpu().numpy(), st.m['item_emb'].cpu().numpy())
                ev = max_norm_err(full_vi.cpu().numpy(), st.v['item_emb'].cpu().numpy())
                if str(e).startswith("('item_emb'") and em <= 1e-5 and ev <= 1e-5:
                    amplified += 1
                    print('OUTLIER (item moments agree: exp_avg %.1e, exp_avg_sq %.1e)' % (em, ev), desc,
                          str(e)[:300].replace('\n', ' '), flush=True)
                else:
                    bad += 1
                    redo = 1
                    print('FAIL (item exp_avg %.1e, exp_avg_sq %.1e)' % (em, ev), desc, str(e)[:300].replace('\n', ' '),
                          flush=True)
        # a failing case once more, step count by step count: where the two runs part (every rank takes part)
        redo_t = torch.tensor([redo if rank == 0 else 0])
        dist.broadcast(redo_t, 0)
        redo = 0
        if int(redo_t.item()):
            for k_steps in range(1, n_steps + 1):
                _, f_u, f_i, _, f_mi, f_vi, _ = run_sharded(k_steps, False)
                if rank == 0:
                    st, tt, _ = run_single(k_steps)
                    worst = int((f_i - tt['item_emb']).abs().argmax().item())
                    worst_u = int((f_u - tt['user_emb']).abs().argmax().item())
                    print('  trace: %2d steps  item_emb %.1e (row %d col %d)  exp_avg %.1e  exp_avg_sq %.1e   user_emb %.1e (row %d '
                          'col %d)' % (
                              k_steps, max_norm_err(f_i.cpu().numpy(), tt['item_emb'].cpu().numpy()), worst // D, worst % D,
                              max_norm_err(f_mi.cpu().numpy(), st.m['item_emb'].cpu().numpy()),
                              max_norm_err(f_vi.cpu().numpy(), st.v['item_emb'].cpu().numpy()),
                              max_norm_err(f_u.cpu().numpy(), tt['user_emb'].cpu().numpy()), worst_u // D, worst_u % D), flush=True)
    if rank == 0:
        print(f'{n} cases at world {world} ({n_eval} with the item-sharded evaluation), {skipped} skipped (refused / overflow), '
              f'{amplified} parameter outliers with agreeing moments, {bad} failures', flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0 and bad:
        sys.exit(1)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(world, port, budget, seed), nprocs=world, join=True)


if __name__ == '__main__':
    main()
