"""Host-side CSR container for user -> item sets (what the device sampler / eval mask / metrics read).

The reference keeps these as scipy CSR matrices (`sampling_matrix`, `exclude_data`, `iteration_matrix`,
data/dataset.py:120-131,174-194).  The kernels need exactly two arrays: indptr int64 [n_rows+1] and
indices int32 [nnz], rows sorted and free of duplicates (scipy's canonical form after coo->csr).
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class UserItemCsr:
    indptr: np.ndarray   # int64 [n_rows + 1]
    indices: np.ndarray  # int32 [nnz], sorted inside each row, no duplicates
    n_rows: int
    n_cols: int

    @staticmethod
    def from_pairs(users, items, n_rows: int, n_cols: int) -> 'UserItemCsr':
        users = np.asarray(users, dtype=np.int64).reshape(-1)
        items = np.asarray(items, dtype=np.int64).reshape(-1)
        if users.shape != items.shape:
            raise ValueError('users / items length mismatch')
        if len(users) and (users.min() < 0 or users.max() >= n_rows or items.min() < 0 or items.max() >= n_cols):
            raise ValueError('interaction outside the [n_users, n_items] grid')
        key = np.unique(users * np.int64(n_cols) + items)
        rows = key // n_cols
        indptr = np.zeros(n_rows + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows, minlength=n_rows), out=indptr[1:])
        return UserItemCsr(indptr, (key % n_cols).astype(np.int32), n_rows, n_cols)

    @property
    def nnz(self) -> int:
        return int(self.indices.shape[0])

    def row(self, r: int) -> np.ndarray:
        return self.indices[self.indptr[r]:self.indptr[r + 1]]

    def row_lengths(self) -> np.ndarray:
        return np.diff(self.indptr)

    def union(self, other: 'UserItemCsr') -> 'UserItemCsr':
        if (self.n_rows, self.n_cols) != (other.n_rows, other.n_cols):
            raise ValueError('shape mismatch')
        ru = np.repeat(np.arange(self.n_rows), self.row_lengths())
        ro = np.repeat(np.arange(other.n_rows), other.row_lengths())
        return UserItemCsr.from_pairs(np.concatenate([ru, ro]), np.concatenate([self.indices, other.indices]),
                                      self.n_rows, self.n_cols)

    def subset_rows(self, first: int, stride: int) -> 'UserItemCsr':
        """Rows first, first+stride, ... as a new CSR (row j of the result = row first + j*stride): the local
        view of a table that is row-sharded by `row % stride == first`."""
        rows = np.arange(first, self.n_rows, stride)
        lens = self.row_lengths()[rows]
        indptr = np.zeros(len(rows) + 1, dtype=np.int64)
        np.cumsum(lens, out=indptr[1:])
        starts = self.indptr[rows]
        idx = np.repeat(starts - indptr[:-1], lens) + np.arange(indptr[-1])
        return UserItemCsr(indptr, self.indices[idx], len(rows), self.n_cols)

    def to_scipy(self, dtype=bool):
        from scipy import sparse as sp
        return sp.csr_matrix((np.ones(self.nnz, dtype=dtype), self.indices, self.indptr),
                             shape=(self.n_rows, self.n_cols))

    def to_device(self, device):
        import torch
        return (torch.from_numpy(self.indptr).to(device), torch.from_numpy(self.indices).to(device))
