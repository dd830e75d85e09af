"""Dense optimiser steps on the HIP device for the autograd path of `Trainer` (train/trainer.py:48-53,147 of the
reference): torch.optim.{Adam, Adagrad, AdamW}.step() re-stated over `hsk_opt_dense`, one launch per parameter.

Same semantics as the torch classes with the arguments the reference passes (lr, weight_decay; everything else
default): parameters without a gradient are skipped, state is created lazily at the first step, `weight_decay` is
decoupled for adamw and L2 for adam / adagrad.  The fused step (`hip_ops.BprMfFusedState`) does not use this class:
it carries the same arithmetic inside its own kernels."""
from typing import Dict, Iterable

import torch

from hassaku_amd import hip_ops


class HipOptimizer:
    def __init__(self, params: Iterable[torch.nn.Parameter], optimizer: str, lr: float, weight_decay: float = 0.0):
        if optimizer not in hip_ops.OPT_KINDS:
            raise ValueError(f'Optimizer {optimizer} not yet implemented')
        self.kind, self.lr, self.wd = optimizer, float(lr), float(weight_decay)
        self.params = [p for p in params if p.requires_grad]
        self.state: Dict[torch.nn.Parameter, dict] = {}

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self):
        for p in self.params:
            if p.grad is None:
                continue
            if p.grad.is_sparse:
                raise RuntimeError('HipOptimizer expects dense gradients')
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = {'step': 0, 'exp_avg': torch.zeros_like(p), 'exp_avg_sq': torch.zeros_like(p)}
            st['step'] += 1
            hip_ops.opt_dense(self.kind, p.data, p.grad.contiguous(), st['exp_avg'], st['exp_avg_sq'], self.lr, self.wd,
                              st['step'])
