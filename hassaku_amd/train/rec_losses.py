"""Recommendation losses with the reference's plugin interface (train/rec_losses.py:10-25,27-145) on the HIP path.

`bpr` is the hot path (hsk_bpr_loss_grad / fused step); `bce` and `sampled_softmax` are the first "next" row of
SURVEY.md section 8(f): same scorer, different epilogue (hsk_rec_loss_grad / fused step with loss_kind).  Every
`compute_loss` is a torch.autograd.Function over one HIP kernel; there is no PyTorch fallback.
"""
import abc
import logging
import math
from enum import Enum

import torch

from hassaku_amd import hip_ops


class RecommenderSystemLoss(abc.ABC):
    def __init__(self):
        super().__init__()
        self.name = 'RecommenderSystemLoss'

    @abc.abstractmethod
    def compute_loss(self, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        ...

    @staticmethod
    @abc.abstractmethod
    def build_from_conf(conf: dict, dataset):
        ...


class _RecLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, kind, log_adjust):
        loss, grad = hip_ops.rec_loss_grad(kind, logits.contiguous(), log_adjust, need_grad=True)
        ctx.save_for_backward(grad)
        return loss.view(())

    @staticmethod
    def backward(ctx, grad_out):
        (grad,) = ctx.saved_tensors
        return grad * grad_out.to(grad.dtype), None, None


def _check_logits(logits):
    if logits.dim() != 2 or logits.shape[1] < 2:
        raise ValueError(f'logits must be [batch, 1 + n_neg], got {tuple(logits.shape)}')


class RecBayesianPersonalizedRankingLoss(RecommenderSystemLoss):
    """mean over (b, n) of -log sigmoid(logits[b,0] - logits[b,1+n]); fp64 scalar like the reference's
    BCEWithLogits on fp64 labels (train/rec_losses.py:68-88)."""
    kind = 'bpr'

    def __init__(self):
        super().__init__()
        self.name = 'RecBayesianPersonalizedRankingLoss'
        logging.info('Built %s (HIP)', self.name)

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return RecBayesianPersonalizedRankingLoss()

    def compute_loss(self, logits: torch.Tensor, labels: torch.Tensor = None) -> torch.Tensor:
        # `labels` (column 0 = 1) only encodes which column is the positive; it is not read.
        _check_logits(logits)
        return _RecLoss.apply(logits, 'bpr', 0.0)


class RecBinaryCrossEntropy(RecommenderSystemLoss):
    """mean over batch*(1+n_neg) of BCEWithLogits(logit, [column == 0]) (train/rec_losses.py:27-53); fp64 scalar."""
    kind = 'bce'

    def __init__(self):
        super().__init__()
        self.name = 'RecBinaryCrossEntropy'
        logging.info('Built %s (HIP)', self.name)

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return RecBinaryCrossEntropy()

    def compute_loss(self, logits: torch.Tensor, labels: torch.Tensor = None) -> torch.Tensor:
        _check_logits(logits)
        return _RecLoss.apply(logits, 'bce', 0.0)


class RecSampledSoftmaxLoss(RecommenderSystemLoss):
    """mean over the batch of -x_pos + logsumexp(x_pos, x_neg + log(n_items / neg_train)); the correction is applied
    only for uniform sampling, exactly like the reference (train/rec_losses.py:91-139).  Unlike the reference the
    caller's logits tensor is not modified in place."""
    kind = 'sampled_softmax'

    def __init__(self, n_items: int = None, train_neg_strategy: str = None, neg_train: int = None):
        super().__init__()
        self.n_items, self.train_neg_strategy, self.neg_train = n_items, train_neg_strategy, neg_train
        self.name = 'RecSampledSoftmaxLoss'
        logging.info('Built %s (HIP)', self.name)

    @property
    def log_adjust(self) -> float:
        if self.train_neg_strategy == 'uniform':
            return math.log(self.n_items / self.neg_train)
        return 0.0

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return RecSampledSoftmaxLoss(n_items=dataset.n_items, train_neg_strategy=conf['train_neg_strategy'],
                                     neg_train=conf['neg_train'])

    def compute_loss(self, logits: torch.Tensor, labels: torch.Tensor = None) -> torch.Tensor:
        _check_logits(logits)
        return _RecLoss.apply(logits, 'sampled_softmax', self.log_adjust)


class RecommenderSystemLossesEnum(Enum):
    bce = RecBinaryCrossEntropy
    bpr = RecBayesianPersonalizedRankingLoss
    sampled_softmax = RecSampledSoftmaxLoss
