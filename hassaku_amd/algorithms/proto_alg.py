"""Anchor / prototype models on the HIP embedding gather -- ACF, UProtoMF, IProtoMF, UIProtoMF
(algorithms/sgd_alg.py:187-570 of the reference; SURVEY.md 8f rank 4: "the prototype models' shared gather").

What these models share with the matrix-factorisation hot path is the embedding gather and its dense backward: every
`nn.Embedding` lookup here goes through `hsk_embedding_gather` / `hsk_embedding_backward` (deterministic, no atomics);
the loss, the optimiser step and the evaluation's top-k / metrics are the same HIP operators the MF model uses
(train/rec_losses.py, train/optim.py, eval/eval.py).  What is specific to them -- a few [*, D] x [D, P] products with
P = 20..100 prototypes, row normalisations, a softmax over P -- is plain dense algebra on small matrices and runs on
the library path (torch ops, i.e. rocBLAS GEMMs + elementwise kernels on the same device and stream).

Same constructors, parameter names / shapes in state_dict(), `build_from_conf` keys, `get_and_reset_other_loss()` keys
and RNG consumption at construction as the reference's classes; `post_val` (the reference's wandb plots of prototype
neighbourhoods, explanations/) is outside the hot path and returns nothing here.
"""
import logging
import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from hassaku_amd import hip_ops
from hassaku_amd.algorithms.base_classes import SGDBasedRecommenderAlgorithm
from hassaku_amd.train.utils import general_weight_init


def cosine_to(x: torch.Tensor, anchors: torch.Tensor, shift: float, lo: float, hi: float) -> torch.Tensor:
    """clamp(shift + <x/|x|, a/|a|>) of every row of x [n, D] against every row of anchors [P, D] -> [n, P]
    (compute_cosine_sim: shift 0, clamp [-1, 1]; compute_shifted_cosine_sim: shift 1, clamp [0, 2])."""
    sim = F.normalize(x) @ F.normalize(anchors).T
    if shift:
        sim = shift + sim
    return torch.clamp(sim, min=lo, max=hi)


def softmax_entropy(p: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
    """Entropy of p = softmax(logits) over the last dim, from the logits (entropy_from_softmax)."""
    return -(p * (logits - torch.logsumexp(logits, dim=-1, keepdim=True))).sum(-1)


class _HipTables(SGDBasedRecommenderAlgorithm):
    """Status word + lookup helper shared by the models below."""

    def __init__(self):
        super().__init__()
        self._status: Optional[torch.Tensor] = None

    def status_word(self) -> torch.Tensor:
        dev = next(self.parameters()).device
        if self._status is None or self._status.device != dev:
            self._status = hip_ops.new_status(dev)
        return self._status

    def check_indices(self):
        if self._status is not None:
            hip_ops.raise_on_status(self._status, self.name)
            self._status.zero_()

    def lookup(self, table: nn.Embedding, idx: torch.Tensor) -> torch.Tensor:
        return hip_ops.embedding(table.weight, idx, self.status_word())

    def post_val(self, curr_epoch: int) -> Dict:
        return {}


def _pairwise_dot(u: torch.Tensor, i: torch.Tensor) -> torch.Tensor:
    """<u[b], i[b, k]> -> [B, K]  (or [B, I] against a shared item list i [I, P] during evaluation)."""
    if i.dim() == 2:
        return u @ i.T
    return (u.unsqueeze(-2) * i).sum(dim=-1)


class ACF(_HipTables):
    """Anchor-based collaborative filtering (algorithms/sgd_alg.py:187-292): users and items are re-expressed as
    softmax-weighted mixtures of `n_anchors` learned anchor vectors; exclusiveness / inclusiveness entropies of the
    items' anchor weights are the extra losses."""

    def __init__(self, n_users: int, n_items: int, embedding_dim: int = 100, n_anchors: int = 20,
                 delta_exc: float = 1e-1, delta_inc: float = 1e-2):
        super().__init__()
        self.n_users, self.n_items = n_users, n_items
        self.embedding_dim, self.n_anchors = embedding_dim, n_anchors
        self.delta_exc, self.delta_inc = delta_exc, delta_inc
        # construction order = the reference's (anchors, users, items): same seed, same initial state
        self.anchors = nn.Parameter(torch.randn([n_anchors, embedding_dim]), requires_grad=True)
        self.user_embed = nn.Embedding(n_users, embedding_dim)
        self.item_embed = nn.Embedding(n_items, embedding_dim)
        self._exc = self._inc = 0
        self.name = 'ACF'
        logging.info('Built %s (HIP gather): %d users, %d items, dim %d, %d anchors', self.name, n_users, n_items,
                     embedding_dim, n_anchors)

    def _mix(self, emb: torch.Tensor):
        logits = emb @ self.anchors.T
        w = torch.softmax(logits, dim=-1)
        return w @ self.anchors, w, logits

    def get_user_representations(self, u_idxs: torch.Tensor) -> torch.Tensor:
        return self._mix(self.lookup(self.user_embed, u_idxs))[0]

    def get_item_representations(self, i_idxs: torch.Tensor) -> Tuple[torch.Tensor, ...]:
        return self._mix(self.lookup(self.item_embed, i_idxs))

    def combine_user_item_representations(self, u_repr, i_repr) -> torch.Tensor:
        return _pairwise_dot(u_repr, i_repr[0])

    def forward(self, u_idxs: torch.Tensor, i_idxs: torch.Tensor) -> torch.Tensor:
        i_repr = self.get_item_representations(i_idxs)
        out = self.combine_user_item_representations(self.get_user_representations(u_idxs), i_repr)
        _, w, logits = i_repr
        self._exc = self._exc + softmax_entropy(w, logits).mean()                 # exclusiveness: peaked per item
        share = w.reshape(-1, self.n_anchors).sum(dim=0) / w.sum()                # inclusiveness: anchors all in use
        self._inc = self._inc + (math.log(self.n_anchors) + (share * torch.log(share)).sum())
        return out

    def get_and_reset_other_loss(self) -> Dict:
        exc, inc = self.delta_exc * self._exc, self.delta_inc * self._inc
        self._exc = self._inc = 0
        return {'reg_loss': exc + inc, 'exc_loss': exc, 'inc_loss': inc}

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return ACF(dataset.n_users, dataset.n_items, conf['embedding_dim'], conf['n_anchors'], conf['delta_exc'],
                   conf['delta_inc'])


class _ProtoSide(_HipTables):
    """One-sided ProtoMF: the `proto` side (users or items) is embedded in D dimensions and represented by its shifted
    cosine similarities to `n_prototypes` prototypes; the other side is a plain P-dimensional embedding."""

    proto_side = 'user'

    def __init__(self, n_users: int, n_items: int, embedding_dim: int = 100, n_prototypes: int = 20,
                 sim_proto_weight: float = 1., sim_batch_weight: float = 1.):
        super().__init__()
        self.n_users, self.n_items = n_users, n_items
        self.embedding_dim, self.n_prototypes = embedding_dim, n_prototypes
        self.sim_proto_weight, self.sim_batch_weight = sim_proto_weight, sim_batch_weight
        users_wide = self.proto_side == 'user'
        # construction order = the reference's: user table, item table, prototypes, then the two re-initialisations
        self.user_embed = nn.Embedding(n_users, embedding_dim if users_wide else n_prototypes)
        self.item_embed = nn.Embedding(n_items, n_prototypes if users_wide else embedding_dim)
        self.prototypes = nn.Parameter(torch.randn([n_prototypes, embedding_dim]) * .1 / embedding_dim,
                                       requires_grad=True)
        self.user_embed.apply(general_weight_init)
        self.item_embed.apply(general_weight_init)
        self._r_proto = self._r_batch = 0
        self.name = 'UProtoMF' if users_wide else 'IProtoMF'
        logging.info('Built %s (HIP gather): %d users, %d items, dim %d, %d prototypes', self.name, n_users, n_items,
                     embedding_dim, n_prototypes)

    def _similarities(self, table: nn.Embedding, idx: torch.Tensor) -> torch.Tensor:
        emb = self.lookup(table, idx)
        sim = cosine_to(emb.reshape(-1, emb.shape[-1]), self.prototypes, 1.0, 0.0, 2.0)
        return sim.reshape(tuple(idx.shape) + (self.n_prototypes,))

    def get_user_representations(self, u_idxs: torch.Tensor) -> torch.Tensor:
        if self.proto_side == 'user':
            return self._similarities(self.user_embed, u_idxs)
        return self.lookup(self.user_embed, u_idxs)

    def get_item_representations(self, i_idxs: torch.Tensor) -> torch.Tensor:
        if self.proto_side == 'item':
            return self._similarities(self.item_embed, i_idxs)
        return self.lookup(self.item_embed, i_idxs)

    def combine_user_item_representations(self, u_repr, i_repr) -> torch.Tensor:
        return _pairwise_dot(u_repr, i_repr)

    def compute_reg_losses(self, sim: torch.Tensor):
        """every prototype should have a close entity in the batch, every entity a close prototype"""
        far = 2 - sim.reshape(-1, self.n_prototypes)
        self._r_proto = self._r_proto + far.min(dim=0).values.mean()
        self._r_batch = self._r_batch + far.min(dim=1).values.mean()

    def forward(self, u_idxs: torch.Tensor, i_idxs: torch.Tensor) -> torch.Tensor:
        u_repr, i_repr = self.get_user_representations(u_idxs), self.get_item_representations(i_idxs)
        self.compute_reg_losses(u_repr if self.proto_side == 'user' else i_repr)
        return self.combine_user_item_representations(u_repr, i_repr)

    def get_and_reset_other_loss(self) -> Dict:
        proto, batch = self.sim_proto_weight * self._r_proto, self.sim_batch_weight * self._r_batch
        self._r_proto = self._r_batch = 0
        return {'reg_loss': proto + batch, 'proto_loss': proto, 'batch_loss': batch}


class UProtoMF(_ProtoSide):
    """ProtoMF with user prototypes (algorithms/sgd_alg.py:295-385)."""
    proto_side = 'user'

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return UProtoMF(dataset.n_users, dataset.n_items, conf['embedding_dim'], conf['n_prototypes'],
                        conf['sim_proto_weight'], conf['sim_batch_weight'])


class IProtoMF(_ProtoSide):
    """ProtoMF with item prototypes (algorithms/sgd_alg.py:388-484)."""
    proto_side = 'item'

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return IProtoMF(dataset.n_users, dataset.n_items, conf['embedding_dim'], conf['n_prototypes'],
                        conf['sim_proto_weight'], conf['sim_batch_weight'])


class UIProtoMF(_HipTables):
    """ProtoMF with user AND item prototypes (algorithms/sgd_alg.py:487-570): the sum of a UProtoMF whose item side is
    a linear projection of the IProtoMF's item embedding, and vice versa."""

    def __init__(self, n_users: int, n_items: int, embedding_dim: int = 100, u_n_prototypes: int = 20,
                 i_n_prototypes: int = 20, u_sim_proto_weight: float = 1., u_sim_batch_weight: float = 1.,
                 i_sim_proto_weight: float = 1., i_sim_batch_weight: float = 1.):
        super().__init__()
        self.n_users, self.n_items, self.embedding_dim = n_users, n_items, embedding_dim
        self.uprotomf = UProtoMF(n_users, n_items, embedding_dim, u_n_prototypes, u_sim_proto_weight, u_sim_batch_weight)
        self.iprotomf = IProtoMF(n_users, n_items, embedding_dim, i_n_prototypes, i_sim_proto_weight, i_sim_batch_weight)
        self.u_to_i_proj = nn.Linear(embedding_dim, i_n_prototypes, bias=False)
        self.i_to_u_proj = nn.Linear(embedding_dim, u_n_prototypes, bias=False)
        self.u_to_i_proj.apply(general_weight_init)
        self.i_to_u_proj.apply(general_weight_init)
        del self.uprotomf.item_embed      # the narrow sides are replaced by the projections
        del self.iprotomf.user_embed
        self.name = 'UIProtoMF'
        logging.info('Built %s (HIP gather)', self.name)

    def get_user_representations(self, u_idxs: torch.Tensor):
        emb = self.lookup(self.uprotomf.user_embed, u_idxs)
        sim = cosine_to(emb, self.uprotomf.prototypes, 1.0, 0.0, 2.0)
        return sim, self.u_to_i_proj(emb)

    def get_item_representations(self, i_idxs: torch.Tensor):
        emb = self.lookup(self.iprotomf.item_embed, i_idxs)
        sim = cosine_to(emb.reshape(-1, emb.shape[-1]), self.iprotomf.prototypes, 1.0, 0.0, 2.0)
        return sim.reshape(tuple(i_idxs.shape) + (sim.shape[-1],)), self.i_to_u_proj(emb)

    def combine_user_item_representations(self, u_repr, i_repr) -> torch.Tensor:
        (u_sim, u_proj), (i_sim, i_proj) = u_repr, i_repr
        return _pairwise_dot(u_sim, i_proj) + _pairwise_dot(u_proj, i_sim)

    def forward(self, u_idxs: torch.Tensor, i_idxs: torch.Tensor) -> torch.Tensor:
        u_repr, i_repr = self.get_user_representations(u_idxs), self.get_item_representations(i_idxs)
        self.uprotomf.compute_reg_losses(u_repr[0])
        self.iprotomf.compute_reg_losses(i_repr[0])
        return self.combine_user_item_representations(u_repr, i_repr)

    def get_and_reset_other_loss(self) -> Dict:
        u = {'user_' + k: v for k, v in self.uprotomf.get_and_reset_other_loss().items()}
        i = {'item_' + k: v for k, v in self.iprotomf.get_and_reset_other_loss().items()}
        return {'reg_loss': u.pop('user_reg_loss') + i.pop('item_reg_loss'), **u, **i}

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return UIProtoMF(dataset.n_users, dataset.n_items, conf['embedding_dim'], conf['u_n_prototypes'],
                         conf['i_n_prototypes'], conf['u_sim_proto_weight'], conf['u_sim_batch_weight'],
                         conf['i_sim_proto_weight'], conf['i_sim_batch_weight'])
