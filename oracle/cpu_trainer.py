"""TEST INFRASTRUCTURE ONLY -- the "Hassaku CPU trainer" yardstick timed by bench.py's cpu_baseline leg.

A restatement, op for op, of what the reference executes per training step on device='cpu'
(train/trainer.py:128-148 with SGDMatrixFactorization, RecBayesianPersonalizedRankingLoss,
torch.optim.AdamW and the numpy rejection sampler of data/dataloader.py:110-128, num_workers=0):
host sampler -> nn.Embedding gathers -> broadcast-mul-sum scorer -> fp64-label BCEWithLogits on
(pos - neg) -> autograd backward (dense embedding grads) -> dense AdamW.  The reference's own
Python cannot travel to the GPU box, hence kind = "port".  Validated against the reference's golden
vectors in tests/test_oracle_golden.py::test_cpu_trainer_port_matches_reference.
"""
import time

import numpy as np
import torch
from torch import nn


class _MF(nn.Module):
    def __init__(self, n_users, n_items, dim, item_bias=True):
        super().__init__()
        self.user_embeddings = nn.Embedding(n_users, dim)
        self.item_embeddings = nn.Embedding(n_items, dim)
        self.item_bias = nn.Embedding(n_items, 1) if item_bias else None
        for e in (self.user_embeddings, self.item_embeddings, self.item_bias):
            if e is not None:
                nn.init.normal_(e.weight, std=0.1 / e.weight.shape[-1])

    def forward(self, u, i):
        ue = self.user_embeddings(u)
        ie = self.item_embeddings(i)
        out = (ue[:, None, :] * ie).sum(dim=-1)
        if self.item_bias is not None:
            out = out + self.item_bias(i).squeeze(-1)
        return out


def _collate(rng, u, pos, indptr, indices, n_items, n_neg):
    B = len(u)
    neg = np.empty((B, n_neg), dtype=np.int64)
    todo_mask = np.ones((B, n_neg), dtype=bool)
    todo = todo_mask.sum()
    while todo:
        neg[todo_mask] = rng.randint(0, high=n_items, size=todo)
        for b in range(B):
            todo_mask[b] = np.isin(neg[b], indices[indptr[u[b]]:indptr[u[b] + 1]], assume_unique=True)
        todo = todo_mask.sum()
    items = np.column_stack([pos, neg]).astype(np.int64)
    labels = np.zeros_like(items, dtype=float)
    labels[:, 0] = 1.0
    return torch.from_numpy(u.astype(np.int64)), torch.from_numpy(items), torch.from_numpy(labels)


def bpr_loss(out, labels):
    diff = out[:, :1] - out[:, 1:]
    tgt = torch.repeat_interleave(labels[:, 0], diff.shape[1])
    return nn.BCEWithLogitsLoss()(diff.flatten(), tgt)


class _PairDataset(torch.utils.data.Dataset):
    """(user, item) of interaction k -- TrainRecDataset.__getitem__ (data/dataset.py:133-140 of the reference)."""

    def __init__(self, coo_user, coo_item):
        self.coo_user, self.coo_item = coo_user, coo_item

    def __len__(self):
        return len(self.coo_user)

    def __getitem__(self, k):
        return self.coo_user[k], self.coo_item[k]


def _loader_collate(batch, indptr, indices, n_items, n_neg):
    """TrainDataLoader._neg_sampling_collate_fn (data/dataloader.py:92-129): runs inside the loader worker, global numpy RNG"""
    u = np.array([b[0] for b in batch], dtype=np.int64)
    pos = np.array([b[1] for b in batch], dtype=np.int64)
    return _collate(np.random, u, pos, indptr, indices, n_items, n_neg)


class CpuTrainer:
    def __init__(self, n_users, n_items, dim, lr, wd, indptr, indices, coo_user, coo_item, n_neg, batch, seed=64,
                 threads=None):
        if threads:
            torch.set_num_threads(threads)
        torch.manual_seed(seed)
        self.rng = np.random.RandomState(seed)
        self.model = _MF(n_users, n_items, dim)
        self.opt = torch.optim.AdamW(self.model.parameters(), lr=lr, weight_decay=wd)
        self.indptr, self.indices = indptr, indices
        self.coo_user, self.coo_item = coo_user, coo_item
        self.n_items, self.n_neg, self.batch = n_items, n_neg, batch

    def step_on(self, u, items, labels):
        out = self.model(u, items)
        loss = bpr_loss(out, labels)
        total = loss + torch.zeros(1)
        val = total.item()
        total.backward()
        self.opt.step()
        self.opt.zero_grad()
        return val

    def step(self):
        sel = self.rng.randint(0, len(self.coo_user), size=self.batch)
        u, items, labels = _collate(self.rng, self.coo_user[sel], self.coo_item[sel], self.indptr, self.indices,
                                    self.n_items, self.n_neg)
        return self.step_on(u, items, labels)

    def time_loader_steps(self, workers=0, warmup=20, steps=200):
        """SURVEY 8(d) protocol: the reference's loader shape -- a shuffling torch DataLoader whose collate_fn draws the
        negatives, `workers` worker processes with prefetch_factor 2 (data/data_utils.py:335-343) -- feeding the same
        step; `warmup` untimed steps, then `steps` timed ones.  -> (steps, seconds)"""
        import functools
        from torch.utils.data import DataLoader
        collate = functools.partial(_loader_collate, indptr=self.indptr, indices=self.indices, n_items=self.n_items,
                                    n_neg=self.n_neg)
        kw = dict(num_workers=workers, prefetch_factor=2, persistent_workers=True) if workers else {}
        loader = DataLoader(_PairDataset(self.coo_user, self.coo_item), batch_size=self.batch, shuffle=True,
                            collate_fn=collate, **kw)
        n, t0, done = 0, None, False
        while not done:
            for u, items, labels in loader:
                if n == warmup:
                    t0 = time.perf_counter()
                self.step_on(u, items, labels)
                n += 1
                if n == warmup + steps:
                    done = True
                    break
        secs = time.perf_counter() - t0
        del loader
        return steps, secs

    def time_steps(self, budget_s=15.0, min_steps=2, max_steps=200):
        """-> (steps, seconds) for a bounded sample of the workload (first step is an untimed warm-up)."""
        self.step()
        n, t0 = 0, time.perf_counter()
        while n < max_steps and (n < min_steps or time.perf_counter() - t0 < budget_s):
            self.step()
            n += 1
        return n, time.perf_counter() - t0
