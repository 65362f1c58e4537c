"""Parameter-holding layers: torch.nn classes (so ``state_dict`` keys, ``load_state_dict``,
``apply(weights_init_normal)`` and optimizers behave exactly as with the reference) whose forward
runs the HIP kernels."""
from __future__ import annotations

import torch
from torch import nn

from . import ops

ACT_NONE, ACT_RELU, ACT_LEAKY = ops.ACT_NONE, ops.ACT_RELU, ops.ACT_LEAKY


class Conv2d(nn.Conv2d):
    """nn.Conv2d, square kernel / stride, zero padding, groups=1, dilation=1 -- every convolution of the reference's
    models/ package is of this form (generator.py:20,34,63,108-110,148,188,214,218,222,228; discriminator.py:14-51,
    62-65); other geometries are refused loudly rather than mis-computed."""

    layer_class = "other"        # config.LAYER_CLASSES entry whose operand type this conv follows (set by the owner)

    def _geometry(self):
        k, s, p = self.kernel_size, self.stride, self.padding
        if k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or self.groups != 1 or self.dilation != (1, 1) \
                or self.padding_mode != "zeros":
            raise NotImplementedError("gan_danet_amd.Conv2d: only square, ungrouped, undilated, zero-padded convs")
        return s[0], p[0]

    def forward(self, x: torch.Tensor, act: int = ACT_NONE) -> torch.Tensor:
        s, p = self._geometry()
        return ops.conv2d(x, self.weight, self.bias, s, p, act, None, self.layer_class)


class BatchNorm2d(nn.BatchNorm2d):
    def forward(self, x: torch.Tensor, act: int = ACT_NONE) -> torch.Tensor:
        training = self.training or not self.track_running_stats
        if self.training and self.track_running_stats:
            self.num_batches_tracked += 1
        mom = 0.1 if self.momentum is None else self.momentum
        return ops.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, training, mom,
                                  self.eps, act)

    def fold_args(self):
        """(gamma, beta, running_mean, running_var, training, momentum, eps) for the fused consumers"""
        training = self.training or not self.track_running_stats
        if self.training and self.track_running_stats:
            self.num_batches_tracked += 1
        mom = 0.1 if self.momentum is None else self.momentum
        return self.weight, self.bias, self.running_mean, self.running_var, training, mom, self.eps


class Linear(nn.Linear):
    def forward(self, x: torch.Tensor, act: int = ACT_NONE) -> torch.Tensor:
        return ops.linear(x, self.weight, self.bias, act)


class LazyLinear(nn.LazyLinear):
    """nn.LazyLinear that materialises into the HIP-backed Linear (same lazy-init semantics and keys)."""
    cls_to_become = Linear

    def initialize_parameters(self, input, *args, **kwargs) -> None:  # tolerate the fused-activation argument
        super().initialize_parameters(input)

    def forward(self, x: torch.Tensor, act: int = ACT_NONE) -> torch.Tensor:  # only before materialisation hooks
        return ops.linear(x, self.weight, self.bias, act)


class ReLU(nn.Module):
    """placeholder at the reference's nn.ReLU index; normally fused into the producer"""

    def __init__(self, inplace: bool = False) -> None:
        super().__init__()

    def forward(self, x):
        return ops.activation(x, ACT_RELU)


class Sigmoid(nn.Module):
    def forward(self, x):
        return ops.activation(x, ops.ACT_SIGMOID)


class LeakyReLU(nn.Module):
    """nn.LeakyReLU(negative_slope): 0.2 (every LeakyReLU of discriminator.py) is the fused-epilogue code, any other
    slope runs the stand-alone kernel"""

    def __init__(self, negative_slope: float = 0.2, inplace: bool = False) -> None:
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, x):
        return ops.leaky_relu(x, self.negative_slope)


class UpsampleBicubic2x(nn.Module):
    """nn.Upsample(scale_factor=2, mode='bicubic', align_corners=False)"""

    def forward(self, x):
        return ops.bicubic_up2(x)
