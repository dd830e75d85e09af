"""Synthetic interaction data in the reference's on-disk layout.

The reference ships no processed dataset (its processors download from the network), so tests and
bench.py use data generated here: Zipf-like item popularity, Poisson user activity, and the
reference's per-user 80/10/10 "temporal" split sizes (n_test = ceil(.1 n), n_val = ceil(.1 n), see
split_temporal_order_ratio_based, data/data_utils.py:241-277).  `write_csv_dataset` emits the five
CSV files RecDataset expects (data/dataset.py:10-23).
"""
import math
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

# shapes named after the BASELINE.json configs (users, items, target number of interactions)
SHAPES = {
    'ml100k': (943, 1682, 55_000),
    'ml1m': (6040, 3706, 575_000),
    'ml10m': (69_878, 10_677, 8_000_000),
    'lfm2b': (16_384, 131_072, 2_000_000),
    'hbm': (262_144, 2_000_000, 10_000_000),   # item table 4.1 GB at D=512: far beyond L2 + Infinity Cache
}


@dataclass
class SyntheticInteractions:
    n_users: int
    n_items: int
    train: np.ndarray  # [n, 2] int64 (user_idx, item_idx), grouped by user
    val: np.ndarray
    test: np.ndarray
    user_group: Optional[np.ndarray] = None  # [n_users] int64 or None


def _sample_user_items(rng, logp, counts, chunk=512):
    """Weighted sampling without replacement for many users at once (Gumbel top-n)."""
    n_items = logp.shape[0]
    out = []
    for lo in range(0, len(counts), chunk):
        c = counts[lo:lo + chunk]
        keys = logp[None, :] + rng.gumbel(size=(len(c), n_items))
        kmax = int(c.max())
        top = np.argpartition(-keys, kmax - 1, axis=1)[:, :kmax]
        # order the kept ones by key so that the first c[i] are the c[i] best
        order = np.argsort(-np.take_along_axis(keys, top, axis=1), axis=1)
        top = np.take_along_axis(top, order, axis=1)
        for i, n in enumerate(c):
            out.append(top[i, :n])
    return out


def generate(n_users: int, n_items: int, n_interactions: int, seed: int = 0, n_groups: int = 0,
             min_per_user: int = 10) -> SyntheticInteractions:
    rng = np.random.default_rng(seed)
    pop = np.arange(1, n_items + 1, dtype=np.float64) ** -0.8
    pop = pop[rng.permutation(n_items)]
    logp = np.log(pop / pop.sum())
    lam = n_interactions / n_users
    counts = rng.poisson(lam, size=n_users)
    counts = np.clip(counts, min_per_user, max(min_per_user, n_items // 2)).astype(np.int64)
    per_user = _sample_user_items(rng, logp, counts)
    tr, va, te = [], [], []
    for u, items in enumerate(per_user):
        items = items[rng.permutation(len(items))]  # the "temporal" order
        n = len(items)
        n_test = math.ceil(n * 0.1)
        n_val = math.ceil(n * 0.1)
        n_train = n - n_val - n_test
        uu = np.full(n, u, dtype=np.int64)
        tr.append(np.stack([uu[:n_train], items[:n_train]], 1))
        va.append(np.stack([uu[n_train:n_train + n_val], items[n_train:n_train + n_val]], 1))
        te.append(np.stack([uu[n - n_test:], items[n - n_test:]], 1))
    group = rng.integers(0, n_groups, size=n_users) if n_groups > 0 else None
    return SyntheticInteractions(n_users, n_items, np.concatenate(tr).astype(np.int64),
                                 np.concatenate(va).astype(np.int64), np.concatenate(te).astype(np.int64), group)


def generate_fast(n_users: int, n_items: int, n_interactions: int, seed: int = 0,
                  min_per_user: int = 10) -> SyntheticInteractions:
    """Vectorised variant for the large bench shapes: items are drawn WITH replacement from the same
    Zipf-like popularity and duplicates are dropped, so users end up with slightly fewer interactions
    than drawn.  Same split rule.  ~2 s for the ml10m shape (8 M interactions)."""
    rng = np.random.default_rng(seed)
    pop = np.arange(1, n_items + 1, dtype=np.float64) ** -0.8
    pop = pop[rng.permutation(n_items)]
    cdf = np.cumsum(pop / pop.sum())
    counts = rng.poisson(n_interactions / n_users, size=n_users)
    counts = np.clip(counts, min_per_user, max(min_per_user, n_items // 2)).astype(np.int64)
    users = np.repeat(np.arange(n_users, dtype=np.int64), counts)
    items = np.minimum(np.searchsorted(cdf, rng.random(len(users))), n_items - 1).astype(np.int64)
    key = np.unique(users * n_items + items)          # drop duplicates; sorted by (user, item)
    key = key[rng.permutation(len(key))]              # random "temporal" order ...
    key = key[np.argsort(key // n_items, kind='stable')]  # ... inside each user
    users, items = key // n_items, key % n_items
    n_u = np.bincount(users, minlength=n_users)
    start = np.concatenate([[0], np.cumsum(n_u)[:-1]])
    pos = np.arange(len(users)) - start[users]
    n_hold = np.ceil(n_u * 0.1).astype(np.int64)      # n_val = n_test = ceil(.1 n)
    n_train = n_u - 2 * n_hold
    is_train = pos < n_train[users]
    is_val = (~is_train) & (pos < (n_train + n_hold)[users])
    is_test = pos >= (n_u - n_hold)[users]
    pairs = np.stack([users, items], 1)
    return SyntheticInteractions(n_users, n_items, pairs[is_train], pairs[is_val], pairs[is_test], None)


def generate_named(name: str, seed: int = 0, n_groups: int = 0, fast: Optional[bool] = None) -> SyntheticInteractions:
    u, i, n = SHAPES[name]
    if fast is None:
        fast = n > 1_000_000
    if fast:
        return generate_fast(u, i, n, seed=seed)
    return generate(u, i, n, seed=seed, n_groups=n_groups)


def write_csv_dataset(data: SyntheticInteractions, path: str):
    """user_idxs.csv, item_idxs.csv, listening_history_{train,val,test}.csv (data/dataset.py:10-23)."""
    import pandas as pd
    os.makedirs(path, exist_ok=True)
    users = pd.DataFrame({'user_idx': np.arange(data.n_users)})
    if data.user_group is not None:
        users['group_idx'] = data.user_group
    users.to_csv(os.path.join(path, 'user_idxs.csv'), index=False)
    pd.DataFrame({'item_idx': np.arange(data.n_items)}).to_csv(os.path.join(path, 'item_idxs.csv'), index=False)
    for split in ('train', 'val', 'test'):
        arr = getattr(data, split)
        pd.DataFrame({'user_idx': arr[:, 0], 'item_idx': arr[:, 1]}).to_csv(
            os.path.join(path, f'listening_history_{split}.csv'), index=False)
    return path


# ------------------------------------------------------------------------------------------------
# interactions generated on the device (BASELINE configs[4]: 100 M users x 10 M items -- no CSV, no host copy)
# ------------------------------------------------------------------------------------------------
DEVICE_DEG_MIN, DEVICE_DEG_SPAN, DEVICE_SKEW = 12, 17, 2      # 12..28 positives per user (mean 20), popular low ids


class DeviceInteractions:
    """Training interactions as the fused step takes them, generated straight into HBM by hsk_synth_* (csrc/
    hsk_synth.hip): csr_indptr int64 [U+1], csr_indices int32 [nnz] (sorted, duplicate-free rows), coo_user int32
    [nnz]; the COO order is the CSR order, so coo_item IS csr_indices.  A pure function of (seed, user id): every rank
    of a job builds identical arrays without a broadcast.  Stands where TrainRecDataset._prepare_data builds the COO /
    CSR matrices from the CSVs (data/dataset.py:120-131 of the reference)."""

    def __init__(self, n_users: int, n_items: int, device, seed: int = 0, deg_min: int = DEVICE_DEG_MIN,
                 deg_span: int = DEVICE_DEG_SPAN, skew: int = DEVICE_SKEW):
        import ctypes  # noqa: F401
        import torch
        from hassaku_amd import _lib
        _lib.require_gpu()
        lib = _lib.load()
        device = torch.device(device)
        self.n_users, self.n_items, self.seed = int(n_users), int(n_items), int(seed) & 0xFFFFFFFFFFFFFFFF
        self.deg_min, self.deg_span, self.skew = int(deg_min), int(deg_span), int(skew)
        stream = torch.cuda.current_stream(device).cuda_stream
        self.csr_indptr = torch.empty(self.n_users + 1, dtype=torch.int64, device=device)
        _lib.check(lib.hsk_synth_degrees(self.n_users, self.deg_min, self.deg_span, self.seed,
                                         self.csr_indptr.data_ptr(), stream), 'hsk_synth_degrees')
        self.csr_indptr.cumsum_(0)                                       # counts -> offsets, in place
        self.nnz = int(self.csr_indptr[-1].item())
        self.csr_indices = torch.empty(self.nnz, dtype=torch.int32, device=device)
        self.coo_user = torch.empty(self.nnz, dtype=torch.int32, device=device)
        _lib.check(lib.hsk_synth_fill(self.n_users, self.n_items, self.deg_min + self.deg_span - 1, self.skew,
                                      self.seed, self.csr_indptr.data_ptr(), self.csr_indices.data_ptr(),
                                      self.coo_user.data_ptr(), stream), 'hsk_synth_fill')
        self.coo_item = self.csr_indices

    def __len__(self):
        return self.nnz

    def device_arrays(self, device=None):
        """The keyword arguments BprMfFusedState / ShardedBprMf take (same keys as TrainRecDataset.device_arrays)."""
        return dict(csr_indptr=self.csr_indptr, csr_indices=self.csr_indices, coo_user=self.coo_user,
                    coo_item=self.coo_item)

    def random_order(self, n: int, seed: int = 64):
        """n interaction positions drawn uniformly (with replacement) -- an epoch's randperm over 2e9 interactions
        would cost 16 GB; a run of steps only ever reads its own stretch of the order."""
        import torch
        gen = torch.Generator(device=self.csr_indptr.device)
        gen.manual_seed(seed)
        return torch.randint(0, self.nnz, (int(n),), dtype=torch.int64, device=self.csr_indptr.device, generator=gen)
