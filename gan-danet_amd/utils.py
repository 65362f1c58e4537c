"""models/utils.py equivalent: ``weights_init_normal``."""
from __future__ import annotations

from torch import nn


def weights_init_normal(module: nn.Module) -> None:
    """Kaiming-normal (fan_in, relu) for conv weights, (1, 0) for BatchNorm, Xavier-normal for linear, zero
    biases (utils.py:7-21).  Initialisation is host-side torch RNG plumbing, as in the reference; ``gamma``
    parameters are untouched (the reference's nn.Parameter arm can never fire under Module.apply)."""
    if isinstance(module, nn.Conv2d):
        nn.init.kaiming_normal_(module.weight, mode="fan_in", nonlinearity="relu")
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)
    elif isinstance(module, nn.BatchNorm2d):
        nn.init.constant_(module.weight, 1)
        nn.init.constant_(module.bias, 0)
    elif isinstance(module, nn.Linear):
        if isinstance(module, nn.modules.lazy.LazyModuleMixin) and module.has_uninitialized_params():
            return  # the recorded reference run only warned here (GAN_DANet_train.ipynb:L398-399): fc1 keeps its default init
        nn.init.xavier_normal_(module.weight)
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)


__all__ = ["weights_init_normal"]
