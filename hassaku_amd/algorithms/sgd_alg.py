"""SGDMatrixFactorization on the HIP path -- drop-in for algorithms/sgd_alg.py:110-184 of the reference.

Same constructor, same parameter names and shapes in state_dict() (user_embeddings.weight [U,D],
item_embeddings.weight [I,D], item_bias.weight [I,1], user_bias.weight [U,1], global_bias [1]), same
initialisation under the same torch seed (tables are created and re-initialised in the reference's
order, so the RNG stream is consumed identically).  The nn.Embedding modules are parameter
containers only: every score is computed by hsk_mf_scores / the eval GEMM of libhassaku_hip.so.
There is no CPU forward -- calling the model on CPU tensors raises.
"""
import logging
import weakref
from typing import NamedTuple, Optional

import torch
from torch import nn

from hassaku_amd import hip_ops
from hassaku_amd.algorithms.base_classes import SGDBasedRecommenderAlgorithm
from hassaku_amd.train.utils import general_weight_init


class UserRepr(NamedTuple):
    """Opaque user representation: the indices whose rows the kernels gather themselves."""
    u_idxs: torch.Tensor


class ItemRepr(NamedTuple):
    i_idxs: torch.Tensor


class _MFScores(torch.autograd.Function):
    """logits = hsk_mf_scores(...); backward = hsk_mf_backward (dense grads, like nn.Embedding(sparse=False))."""

    @staticmethod
    def forward(ctx, user_emb, item_emb, item_bias, user_bias, global_bias, u_idxs, i_idxs, status):
        ib = None if item_bias is None else item_bias.view(-1)
        ub = None if user_bias is None else user_bias.view(-1)
        out = hip_ops.mf_scores(user_emb, item_emb, ib, ub, global_bias, u_idxs, i_idxs, status)
        ctx.save_for_backward(user_emb, item_emb, u_idxs, i_idxs)
        ctx.has = (item_bias is not None, user_bias is not None, global_bias is not None)
        ctx.status = status
        return out

    @staticmethod
    def backward(ctx, grad_out):
        user_emb, item_emb, u_idxs, i_idxs = ctx.saved_tensors
        has_ib, has_ub, has_gb = ctx.has
        g_u, g_i, g_ib, g_ub, g_gb = hip_ops.mf_backward(user_emb, item_emb, u_idxs, i_idxs, grad_out.contiguous(),
                                                         has_ib, has_ub, has_gb, ctx.status)
        return (g_u, g_i, None if g_ib is None else g_ib.view(-1, 1), None if g_ub is None else g_ub.view(-1, 1),
                g_gb, None, None, None)


class _BiasScores(torch.autograd.Function):
    """logits = ub[u] + ib[i] + gb through hsk_mf_scores(dim=0); backward = hsk_mf_backward(dim=0)."""

    @staticmethod
    def forward(ctx, item_bias, user_bias, global_bias, u_idxs, i_idxs, status):
        out = hip_ops.bias_scores(item_bias.view(-1), user_bias.view(-1), global_bias, u_idxs, i_idxs, status)
        ctx.save_for_backward(u_idxs, i_idxs)
        ctx.sizes = (user_bias.shape[0], item_bias.shape[0])
        ctx.status = status
        return out

    @staticmethod
    def backward(ctx, grad_out):
        u_idxs, i_idxs = ctx.saved_tensors
        n_users, n_items = ctx.sizes
        g_ib, g_ub, g_gb = hip_ops.bias_backward(n_users, n_items, u_idxs, i_idxs, grad_out.contiguous(), True,
                                                 ctx.status)
        return g_ib.view(-1, 1), g_ub.view(-1, 1), g_gb, None, None, None


class SGDBaseline(SGDBasedRecommenderAlgorithm):
    """Global + user + item bias (algorithms/sgd_alg.py:72-107 of the reference): same constructor, parameter names
    (user_bias.weight [U,1], item_bias.weight [I,1], global_bias [1]) and initialisation order."""

    def __init__(self, n_users: int, n_items: int):
        super().__init__()
        self.n_users, self.n_items = n_users, n_items
        self.user_bias = nn.Embedding(n_users, 1)
        self.item_bias = nn.Embedding(n_items, 1)
        self.global_bias = nn.Parameter(torch.zeros(1), requires_grad=True)
        self.apply(general_weight_init)
        self.name = 'SGDBaseline'
        self._status: Optional[torch.Tensor] = None
        logging.info('Built %s (HIP)', self.name)

    def status_word(self) -> torch.Tensor:
        dev = self.item_bias.weight.device
        if self._status is None or self._status.device != dev:
            self._status = hip_ops.new_status(dev)
        return self._status

    def check_indices(self):
        if self._status is not None:
            hip_ops.raise_on_status(self._status, self.name)
            self._status.zero_()

    def get_user_representations(self, u_idxs: torch.Tensor) -> UserRepr:
        return UserRepr(u_idxs)

    def get_item_representations(self, i_idxs: torch.Tensor) -> ItemRepr:
        return ItemRepr(i_idxs)

    def combine_user_item_representations(self, u_repr: UserRepr, i_repr: ItemRepr) -> torch.Tensor:
        u_idxs, i_idxs = u_repr.u_idxs, i_repr.i_idxs
        if i_idxs.dim() == 1:   # evaluation form (eval/eval.py:240-248): every user against the item list
            i_idxs = i_idxs.unsqueeze(0).expand(u_idxs.numel(), -1)
        return _BiasScores.apply(self.item_bias.weight, self.user_bias.weight, self.global_bias, u_idxs.contiguous(),
                                 i_idxs.contiguous(), self.status_word())

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return SGDBaseline(dataset.n_users, dataset.n_items)


class SGDMatrixFactorization(SGDBasedRecommenderAlgorithm):
    """Matrix factorisation scored by dot product (+ optional user / item / global bias)."""

    def __init__(self, n_users: int, n_items: int, embedding_dim: int = 100, use_user_bias: bool = False,
                 use_item_bias: bool = False, use_global_bias: bool = False):
        super().__init__()
        self.n_users, self.n_items, self.embedding_dim = n_users, n_items, embedding_dim
        self.use_user_bias, self.use_item_bias, self.use_global_bias = use_user_bias, use_item_bias, use_global_bias

        self.user_embeddings = nn.Embedding(n_users, embedding_dim)
        self.item_embeddings = nn.Embedding(n_items, embedding_dim)
        if use_user_bias:
            self.user_bias = nn.Embedding(n_users, 1)
        if use_item_bias:
            self.item_bias = nn.Embedding(n_items, 1)
        self.apply(general_weight_init)
        if use_global_bias:
            self.global_bias = nn.Parameter(torch.zeros(1))

        self.name = 'SGDMatrixFactorization'
        self._status: Optional[torch.Tensor] = None
        logging.info('Built %s (HIP) dim=%d user_bias=%s item_bias=%s global_bias=%s', self.name, embedding_dim,
                     use_user_bias, use_item_bias, use_global_bias)

    # -- parameter access for the fused trainer / evaluator ------------------------------------
    def tables(self):
        """(user_emb [U,D], item_emb [I,D], item_bias [I]|None, user_bias [U]|None, global_bias [1]|None) views."""
        ib = self.item_bias.weight.data.view(-1) if self.use_item_bias else None
        ub = self.user_bias.weight.data.view(-1) if self.use_user_bias else None
        gb = self.global_bias.data if self.use_global_bias else None
        return self.user_embeddings.weight.data, self.item_embeddings.weight.data, ib, ub, gb

    def status_word(self) -> torch.Tensor:
        dev = self.user_embeddings.weight.device
        if self._status is None or self._status.device != dev:
            self._status = hip_ops.new_status(dev)
        return self._status

    def check_indices(self):
        """Raise IndexError if a kernel saw an out-of-range index since the last check (one host sync)."""
        if self._status is not None:
            hip_ops.raise_on_status(self._status, self.name)
            self._status.zero_()

    # -- plugin surface ------------------------------------------------------------------------
    def get_user_representations(self, u_idxs: torch.Tensor) -> UserRepr:
        return UserRepr(u_idxs)

    def get_item_representations(self, i_idxs: torch.Tensor) -> ItemRepr:
        return ItemRepr(i_idxs)

    def combine_user_item_representations(self, u_repr: UserRepr, i_repr: ItemRepr) -> torch.Tensor:
        u_idxs, i_idxs = u_repr.u_idxs, i_repr.i_idxs
        if i_idxs.dim() == 1:
            # evaluation form (eval/eval.py:240-248): every user against the item list i_idxs
            return self._score_all(u_idxs, i_idxs)
        w = self.user_embeddings.weight
        return _MFScores.apply(w, self.item_embeddings.weight,
                               self.item_bias.weight if self.use_item_bias else None,
                               self.user_bias.weight if self.use_user_bias else None,
                               self.global_bias if self.use_global_bias else None,
                               u_idxs.contiguous(), i_idxs.contiguous(), self.status_word())

    @torch.no_grad()
    def _score_all(self, u_idxs: torch.Tensor, i_idxs: torch.Tensor) -> torch.Tensor:
        n = i_idxs.numel()
        # whole catalogue in catalogue order?  Decided by comparing with arange (any permutation takes the general
        # path).  The verdict is remembered for THE TENSOR OBJECT it was reached on (held by a weak reference and
        # compared with `is`, plus its in-place version counter), so an evaluation loop that passes the same index
        # tensor every batch syncs once -- never keyed on the address: the caching allocator hands a freed block to
        # the next tensor of the same size.
        cached = getattr(self, '_full_ref', None)
        if cached is None or cached[0]() is not i_idxs or cached[1] != i_idxs._version:
            full = n == self.n_items and bool(
                torch.equal(i_idxs, torch.arange(n, dtype=i_idxs.dtype, device=i_idxs.device)))
            self._full_ref = (weakref.ref(i_idxs), i_idxs._version, full)
        full = self._full_ref[2]
        user_emb, item_emb, ib, ub, gb = self.tables()
        if full:
            _, _, scores = hip_ops.mf_eval_topk(user_emb, item_emb, ib, ub, gb, u_idxs.contiguous(), 0,
                                                status=self.status_word())
            return scores
        return hip_ops.mf_scores(user_emb, item_emb, ib, ub, gb, u_idxs.contiguous(),
                                 i_idxs.unsqueeze(0).expand(u_idxs.numel(), -1).contiguous(), self.status_word())

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return SGDMatrixFactorization(dataset.n_users, dataset.n_items, conf['embedding_dim'], conf['use_user_bias'],
                                      conf['use_item_bias'], conf['use_global_bias'])
