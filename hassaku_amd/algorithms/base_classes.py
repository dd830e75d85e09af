"""Plugin base classes with the reference's method names (algorithms/base_classes.py:12-52,88-165).

Only the SGD family is in scope; the sparse-matrix family keeps its abstract
interface so that out-of-scope algorithms written against it still type-check.
"""
import abc
import logging
import os
from typing import Dict

import torch
from torch import nn


class RecommenderAlgorithm(abc.ABC):
    """predict(u_idxs [B], i_idxs [B, n]) -> scores [B, n]; save/load; build_from_conf(conf, dataset)."""

    def __init__(self):
        super().__init__()
        self.name = 'RecommenderAlgorithm'

    @abc.abstractmethod
    def predict(self, u_idxs: torch.Tensor, i_idxs: torch.Tensor) -> torch.Tensor:
        ...

    @abc.abstractmethod
    def save_model_to_path(self, path: str):
        ...

    @abc.abstractmethod
    def load_model_from_path(self, path: str):
        ...

    @staticmethod
    @abc.abstractmethod
    def build_from_conf(conf: dict, dataset):
        ...


class SGDBasedRecommenderAlgorithm(RecommenderAlgorithm, nn.Module):
    """Models trained by mini-batch SGD through `Trainer` (algorithms/base_classes.py:88-165)."""

    def __init__(self):
        super().__init__()
        self.name = 'SGDBasedRecommenderAlgorithm'

    def forward(self, u_idxs: torch.Tensor, i_idxs: torch.Tensor) -> torch.Tensor:
        return self.combine_user_item_representations(self.get_user_representations(u_idxs),
                                                      self.get_item_representations(i_idxs))

    @abc.abstractmethod
    def get_user_representations(self, u_idxs: torch.Tensor):
        ...

    @abc.abstractmethod
    def get_item_representations(self, i_idxs: torch.Tensor):
        ...

    @abc.abstractmethod
    def combine_user_item_representations(self, u_repr, i_repr) -> torch.Tensor:
        ...

    def get_and_reset_other_loss(self) -> Dict:
        """At least {'reg_loss': tensor[1]}; MF has no extra loss (algorithms/base_classes.py:139-148)."""
        return {'reg_loss': torch.zeros(1)}

    @torch.no_grad()
    def predict(self, u_idxs: torch.Tensor, i_idxs: torch.Tensor) -> torch.Tensor:
        self.eval()
        return self(u_idxs, i_idxs)

    def save_model_to_path(self, path: str):
        hook = getattr(self, '_pre_save_hook', None)   # a fused trainer registers its flush(): no lazily updated row
        if callable(hook):                             # reaches the checkpoint with pending zero-gradient steps
            hook()
        torch.save(self.state_dict(), os.path.join(path, 'model.pth'))
        logging.info('Model Saved')

    def load_model_from_path(self, path: str):
        device = next(self.parameters()).device
        self.load_state_dict(torch.load(os.path.join(path, 'model.pth'), map_location=device))
        logging.info('Model Loaded')
