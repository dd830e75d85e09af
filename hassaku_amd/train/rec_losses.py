"""Recommendation losses with the reference's plugin interface (train/rec_losses.py:10-25,56-88,142-145).

In scope: `bpr` on the HIP path (hsk_bpr_loss_grad).  `bce` and `sampled_softmax` are the next row of
SURVEY.md section 8(f) and are not built yet: selecting them raises NotImplementedError instead of
silently falling back to PyTorch.
"""
import abc
import logging
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


class _BprLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits):
        loss, grad = hip_ops.bpr_loss_grad(logits.contiguous(), need_grad=True)
        ctx.save_for_backward(grad)
        return loss.view(())

    @staticmethod
    def backward(ctx, grad_out):
        (grad,) = ctx.saved_tensors
        return grad * grad_out.to(grad.dtype)


class RecBayesianPersonalizedRankingLoss(RecommenderSystemLoss):
    """mean over (b, n) of -log sigmoid(logits[b,0] - logits[b,1+n]); fp64 scalar like the reference's
    BCEWithLogits on fp64 labels (train/rec_losses.py:68-88)."""

    def __init__(self):
        super().__init__()
        self.name = 'RecBayesianPersonalizedRankingLoss'
        logging.info('Built %s (HIP)', self.name)

    @staticmethod
    def build_from_conf(conf: dict, dataset):
        return RecBayesianPersonalizedRankingLoss()

    def compute_loss(self, logits: torch.Tensor, labels: torch.Tensor = None) -> torch.Tensor:
        # `labels` (column 0 = 1) only encodes which column is the positive; it is not read.
        if logits.dim() != 2 or logits.shape[1] < 2:
            raise ValueError(f'logits must be [batch, 1 + n_neg], got {tuple(logits.shape)}')
        return _BprLoss.apply(logits)


class _NotBuiltYet(RecommenderSystemLoss):
    tag = ''

    @classmethod
    def build_from_conf(cls, conf: dict, dataset):
        raise NotImplementedError(f"rec_loss '{cls.tag}' is not on the HIP path yet (SURVEY.md 8f, next tier); "
                                  f"use rec_loss: bpr")

    def compute_loss(self, logits, labels):
        raise NotImplementedError(self.tag)


class RecBinaryCrossEntropy(_NotBuiltYet):
    tag = 'bce'


class RecSampledSoftmaxLoss(_NotBuiltYet):
    tag = 'sampled_softmax'


class RecommenderSystemLossesEnum(Enum):
    bce = RecBinaryCrossEntropy
    bpr = RecBayesianPersonalizedRankingLoss
    sampled_softmax = RecSampledSoftmaxLoss
