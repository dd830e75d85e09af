"""Training loader with on-device negative sampling -- drop-in for data/dataloader.py:17-129.

Strategies: 'uniform' and 'popular' (alias table over pop_distribution ** squashing factor).  The reference draws
negatives on the host inside the DataLoader's collate_fn (numpy randint + a per-row
scipy CSR slice + np.isin, repeated until nothing collides) and ships int64/float64 tensors to the
device every step.  Here the epoch permutation, the positives and the negatives all live in HBM:
`hsk_sample_negatives_uniform` (Philox4x32-10, rejection against the user's sorted CSR row) produces the
same law -- i.i.d. uniform over the items the user has not interacted with, with replacement.

Two ways to consume it:
  * iterate it like the reference's loader: yields (u_idxs i64 [B], i_idxs i64 [B,1+N], labels f64 [B,1+N])
    as DEVICE tensors (column 0 positive, labels[:,0] = 1);
  * `fused_batches()`: yields (order, start, batch) descriptors for `hsk_bprmf_train_step_sampled`, which
    samples inside the fused step and never materialises the batch.
"""
import logging
from typing import Iterator, Optional, Tuple

import torch

from hassaku_amd import hip_ops
from hassaku_amd.data.dataset import TrainRecDataset


class InteractionSampler:
    pass


class NegativeSampler(InteractionSampler):
    """Parameters of the negative sampling (data/dataloader.py:17-54)."""

    def __init__(self, train_dataset: TrainRecDataset, n_neg: int = 10, neg_sampling_strategy: str = 'uniform',
                 squashing_factor_pop_sampling: float = 1.):
        assert n_neg > 0, 'Number of negatives should be > 0!'
        assert neg_sampling_strategy in ['uniform', 'popular'], \
            f'<{neg_sampling_strategy}> is not a valid negative sampling strategy!'
        assert squashing_factor_pop_sampling >= 0, 'Squashing factor for popularity sampling should be positive!'
        self.dataset = train_dataset
        self.n_neg = n_neg
        self.neg_sampling_strategy = neg_sampling_strategy
        self.squashing_factor_pop_sampling = squashing_factor_pop_sampling
        self.n_items = train_dataset.n_items
        self.pop_distribution = train_dataset.pop_distribution.copy()
        self.name = 'NegativeSampler'
        self._alias_host = None
        self._alias_dev = {}
        logging.info('Built %s n_neg=%d strategy=%s', self.name, n_neg, neg_sampling_strategy)

    def alias(self, device):
        """Device alias table (prob, idx) of pop_distribution ** squashing_factor for 'popular' sampling
        (data/dataloader.py:59-64 of the reference: p = pop^alpha / sum), None for 'uniform'."""
        if self.neg_sampling_strategy != 'popular':
            return None
        if self._alias_host is None:
            import numpy as np
            p = np.power(self.pop_distribution, self.squashing_factor_pop_sampling)
            self._alias_host = hip_ops.build_alias_table(p / p.sum())
        key = str(device)
        if key not in self._alias_dev:
            prob, idx = self._alias_host
            self._alias_dev[key] = (torch.from_numpy(prob).to(device), torch.from_numpy(idx).to(device))
        return self._alias_dev[key]


class TrainDataLoader:
    """Batches of positives + sampled negatives.  Constructor arguments follow the reference's
    TrainDataLoader(interaction_sampler, dataset, batch_size, shuffle, ...) (data/dataloader.py:72-90);
    worker / pin-memory options are accepted and ignored (nothing runs on the host)."""

    def __init__(self, interaction_sampler: InteractionSampler, dataset: TrainRecDataset, batch_size: Optional[int] = 1,
                 shuffle: bool = False, num_workers: int = 0, drop_last: bool = False, device='cuda',
                 seed: Optional[int] = None, **_ignored):
        if not isinstance(interaction_sampler, NegativeSampler):
            raise ValueError('Invalid Interaction Sampler')
        self.interaction_sampler = interaction_sampler
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle, self.drop_last = shuffle, drop_last
        self.device = torch.device(device)
        # like the reference's RandomSampler, the shuffle is seeded from torch's global RNG (reproducible(seed))
        self.seed = int(torch.initial_seed() & 0x7fffffffffffffff) if seed is None else int(seed)
        self.epoch = 0
        self._gen: Optional[torch.Generator] = None
        self._labels = {}

    def __len__(self) -> int:
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    # -- epoch order ----------------------------------------------------------------------------
    def _epoch_order(self) -> Optional[torch.Tensor]:
        if not self.shuffle:
            return None
        if self._gen is None:
            self._gen = torch.Generator(device=self.device)
            self._gen.manual_seed(self.seed)
        return torch.randperm(len(self.dataset), device=self.device, generator=self._gen)

    def fused_batches(self) -> Iterator[Tuple[Optional[torch.Tensor], int, int]]:
        """(order, start, batch) per step of one epoch; negatives are drawn inside the fused step."""
        order = self._epoch_order()
        self.epoch += 1
        n, bs = len(self.dataset), self.batch_size
        for b in range(len(self)):
            yield order, b * bs, min(bs, n - b * bs)

    def _label_tensor(self, rows: int, cols: int) -> torch.Tensor:
        key = (rows, cols)
        if key not in self._labels:
            lab = torch.zeros((rows, cols), dtype=torch.float64, device=self.device)
            lab[:, 0] = 1.
            self._labels[key] = lab
        return self._labels[key]

    def __iter__(self):
        arrays = self.dataset.device_arrays(self.device)
        n_neg, n_items = self.interaction_sampler.n_neg, self.interaction_sampler.n_items
        status = hip_ops.new_status(self.device)
        epoch = self.epoch
        for step, (order, start, nb) in enumerate(self.fused_batches()):
            sel = order[start:start + nb] if order is not None else torch.arange(start, start + nb, device=self.device)
            u = arrays['coo_user'][sel].to(torch.int64)
            pos = arrays['coo_item'][sel].to(torch.int64)
            neg = hip_ops.sample_negatives_uniform(arrays['csr_indptr'], arrays['csr_indices'], n_items, u, n_neg,
                                                   seed=self.seed, stream_id=(epoch << 32) | step, status=status,
                                                   alias=self.interaction_sampler.alias(self.device))
            items = torch.cat([pos[:, None], neg], dim=1)
            yield u, items, self._label_tensor(nb, 1 + n_neg)
        hip_ops.raise_on_status(status, 'TrainDataLoader')
