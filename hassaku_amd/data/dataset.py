"""Datasets in the reference's 5-CSV layout (data/dataset.py:10-23), kept in the form the kernels read.

A processed dataset directory holds user_idxs.csv (`user_idx`, optional `group_idx`), item_idxs.csv
(`item_idx`) and listening_history_{train,val,test}.csv (`user_idx`, `item_idx`).  Where the reference
builds scipy COO/CSR matrices and densifies label rows per user on the host, these classes build once:
  * the interaction list (COO order of the CSV, duplicates kept -- `len()` is the number of CSV rows,
    data/dataset.py:133-134),
  * a sorted, duplicate-free user->items CSR (sampling / exclusion / ground truth),
and hand both to the device on request (`device_arrays`).  Attribute names the Trainer / evaluator of
the reference touch (`n_users`, `n_items`, `n_user_groups`, `user_to_user_group`, `pop_distribution`,
`exclude_data`, `iteration_matrix`, `sampling_matrix`) are all present.
"""
import logging
import os
from typing import Dict, Optional

import numpy as np
import pandas as pd
import torch
from torch.utils import data

from hassaku_amd.data.csr import UserItemCsr

SPLITS = ('train', 'val', 'test')


def _read_pairs(data_path: str, split: str) -> np.ndarray:
    frame = pd.read_csv(os.path.join(data_path, f'listening_history_{split}.csv'), usecols=['user_idx', 'item_idx'])
    return frame[['user_idx', 'item_idx']].to_numpy(dtype=np.int64)


class RecDataset(data.Dataset):
    """Common part: sizes, optional user groups, the split's interaction pairs (data/dataset.py:26-86)."""

    def __init__(self, data_path: str, split_set: str):
        if split_set not in SPLITS:
            raise AssertionError(f'<{split_set}> is not a valid value for split set!')
        self.data_path, self.split_set = data_path, split_set
        users = pd.read_csv(os.path.join(data_path, 'user_idxs.csv'))
        items = pd.read_csv(os.path.join(data_path, 'item_idxs.csv'))
        self.n_users, self.n_items = len(users), len(items)
        self.user_to_user_group: Optional[torch.Tensor] = None
        self.n_user_groups = 0
        if 'group_idx' in users.columns:
            by_user = users.sort_values('user_idx')
            self.user_to_user_group = torch.tensor(by_user['group_idx'].to_numpy(), dtype=torch.float32)
            self.n_user_groups = int(users['group_idx'].nunique())
        self.pairs = _read_pairs(data_path, split_set)
        self.name = 'RecDataset'
        self._device_cache: Dict[str, Dict[str, torch.Tensor]] = {}
        logging.info('Built %s: %s/%s users=%d items=%d interactions=%d groups=%d', self.name, data_path, split_set,
                     self.n_users, self.n_items, len(self.pairs), self.n_user_groups)

    def __len__(self):
        raise NotImplementedError('use TrainRecDataset for training or FullEvalDataset for evaluation')

    def __getitem__(self, index):
        raise NotImplementedError('use TrainRecDataset for training or FullEvalDataset for evaluation')


class TrainRecDataset(RecDataset):
    """Iterates over the positive interactions; owns the per-user positive sets the sampler rejects
    against and the item popularity distribution (data/dataset.py:89-140)."""

    def __init__(self, data_path: str, delete_lhs: bool = True):
        super().__init__(data_path, 'train')
        self.delete_lhs = delete_lhs
        self.coo_user = np.ascontiguousarray(self.pairs[:, 0], dtype=np.int32)
        self.coo_item = np.ascontiguousarray(self.pairs[:, 1], dtype=np.int32)
        self.sampling_csr = UserItemCsr.from_pairs(self.pairs[:, 0], self.pairs[:, 1], self.n_users, self.n_items)
        popularity = np.bincount(self.pairs[:, 1], minlength=self.n_items).astype(np.float64)
        self.pop_distribution = popularity / popularity.sum()
        self.name = 'TrainRecDataset'

    # scipy views for code written against the reference's attributes
    @property
    def iteration_matrix(self):
        from scipy import sparse as sp
        return sp.coo_matrix((np.ones(len(self.coo_user), dtype=np.int16), (self.coo_user, self.coo_item)),
                             shape=(self.n_users, self.n_items))

    @property
    def sampling_matrix(self):
        return self.sampling_csr.to_scipy(np.int16)

    def __len__(self):
        return len(self.coo_user)

    def __getitem__(self, index):
        return np.int64(self.coo_user[index]), np.int64(self.coo_item[index]), 1.

    def device_arrays(self, device) -> Dict[str, torch.Tensor]:
        key = str(device)
        if key not in self._device_cache:
            indptr, indices = self.sampling_csr.to_device(device)
            self._device_cache[key] = {
                'csr_indptr': indptr, 'csr_indices': indices,
                'coo_user': torch.from_numpy(self.coo_user).to(device),
                'coo_item': torch.from_numpy(self.coo_item).to(device)}
        return self._device_cache[key]


class FullEvalDataset(RecDataset):
    """All users against all items.  Ground truth = the split's interactions; excluded from ranking =
    train (val split) or train + val (test split) (data/dataset.py:143-201)."""

    def __init__(self, data_path: str, split_set: str, delete_lhs: bool = True):
        if split_set not in ('val', 'test'):
            raise AssertionError(f'<{split_set}> is not a valid evaluation split!')
        super().__init__(data_path, split_set)
        self.delete_lhs = delete_lhs
        self.label_csr = UserItemCsr.from_pairs(self.pairs[:, 0], self.pairs[:, 1], self.n_users, self.n_items)
        excl = _read_pairs(data_path, 'train')
        if split_set == 'test':
            excl = np.concatenate([excl, _read_pairs(data_path, 'val')])
        self.exclude_csr = UserItemCsr.from_pairs(excl[:, 0], excl[:, 1], self.n_users, self.n_items)
        self.name = 'FullEvalDataset'

    @property
    def exclude_data(self):
        return self.exclude_csr.to_scipy(bool)

    @property
    def iteration_matrix(self):
        return self.label_csr.to_scipy(np.int16)

    def __len__(self):
        return self.n_users

    def __getitem__(self, user_index):
        """(user, arange(n_items), dense float32 label row) -- the reference's contract for generic loaders
        (data/dataset.py:199-201).  The HIP evaluator never calls this: it reads the CSRs on the device."""
        labels = np.zeros(self.n_items, dtype=np.float32)
        labels[self.label_csr.row(user_index)] = 1.
        return user_index, np.arange(self.n_items), labels

    def device_arrays(self, device) -> Dict[str, torch.Tensor]:
        key = str(device)
        if key not in self._device_cache:
            lp, li = self.label_csr.to_device(device)
            ep, ei = self.exclude_csr.to_device(device)
            self._device_cache[key] = {'label_indptr': lp, 'label_indices': li, 'excl_indptr': ep, 'excl_indices': ei}
        return self._device_cache[key]
