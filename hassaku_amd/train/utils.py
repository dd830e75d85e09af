"""Weight initialisation of the reference (train/utils.py:5-13): embedding tables ~ N(0, (0.1/last_dim)^2)."""
import torch
from torch import nn


def general_weight_init(m: nn.Module):
    if type(m) is nn.Embedding and m.weight.requires_grad:
        torch.nn.init.normal_(m.weight, std=0.1 / m.weight.shape[-1])
    elif type(m) is nn.Linear and m.weight.requires_grad:
        torch.nn.init.kaiming_uniform_(m.weight, nonlinearity='relu')
        if m.bias is not None and m.bias.requires_grad:
            torch.nn.init.zeros_(m.bias)
