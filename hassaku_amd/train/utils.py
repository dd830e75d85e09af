"""Parameter initialisation used by `Module.apply` in the model constructors.

Must consume the torch RNG exactly as the reference does (train/utils.py:5-13 there) so that the same seed gives
bit-identical initial tables (tests/test_host_logic.py::test_state_dict_keys_shapes_and_init_match_reference):
embedding tables ~ N(0, (0.1 / last_dim)^2); linear layers Kaiming-uniform (relu gain) with zero bias; anything else,
and frozen parameters, untouched."""
import torch
from torch import nn


def _embedding_table(layer: nn.Embedding) -> None:
    table = layer.weight
    if table.requires_grad:
        nn.init.normal_(table, mean=0.0, std=0.1 / table.shape[-1])


def _linear_layer(layer: nn.Linear) -> None:
    if layer.weight.requires_grad:
        nn.init.kaiming_uniform_(layer.weight, nonlinearity='relu')
        if layer.bias is not None and layer.bias.requires_grad:
            nn.init.zeros_(layer.bias)


_BY_TYPE = {nn.Embedding: _embedding_table, nn.Linear: _linear_layer}


def general_weight_init(m: nn.Module) -> None:
    init = _BY_TYPE.get(type(m))      # exact type match, like the reference: subclasses are left alone
    if init is not None:
        with torch.no_grad():
            init(m)
