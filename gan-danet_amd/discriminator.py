"""Discriminators of GAN-DANet on HIP kernels (models/discriminator.py): same class names, constructor
signatures and ``state_dict`` keys.  LeakyReLU(0.2) is fused into each conv / linear epilogue."""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .layers import ACT_LEAKY, BatchNorm2d, Conv2d, LazyLinear, LeakyReLU, Linear


class Discriminator1(nn.Module):
    """discriminator.py:57-77: 4 x (conv3x3 stride 2 + LeakyReLU) -> flatten -> LazyLinear(1024) -> LeakyReLU
    -> Linear(1024, 1).  fc1 holds 512*(H/16)*(W/16)*1024 weights: an HBM-streaming skinny GEMM."""

    def __init__(self, input_channels: int = 1) -> None:
        super().__init__()
        self.conv1 = Conv2d(input_channels, 64, kernel_size=3, stride=2, padding=1)
        self.conv2 = Conv2d(64, 128, kernel_size=3, stride=2, padding=1)
        self.conv3 = Conv2d(128, 256, kernel_size=3, stride=2, padding=1)
        self.conv4 = Conv2d(256, 512, kernel_size=3, stride=2, padding=1)
        self.fc1 = LazyLinear(1024)
        self.fc2 = Linear(1024, 1)
        self.activation = LeakyReLU(negative_slope=0.2, inplace=True)
        for m in (self.conv1, self.conv2, self.conv3, self.conv4):
            m.layer_class = "disc"

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        convs = (self.conv1, self.conv2, self.conv3, self.conv4)
        if ops.disc1_trunk_eligible(x, [c.weight for c in convs]):
            # one node on pixel-major bf16 activations (ops.Disc1TrunkFn): forward, data and weight gradients
            x = ops.Disc1TrunkFn.apply(x, *[t for c in convs for t in (c.weight, c.bias)])
        else:
            for conv in convs:
                x = conv(x, ACT_LEAKY)
            x = x.flatten(1)
        x = self.fc1(x, ACT_LEAKY)
        return self.fc2(x)


class SRGAND(nn.Module):
    """discriminator.py:8-54.  Exported by the reference but never built by the train loop:
    parameters/keys are provided for checkpoint compatibility; the 4x4 stride-2 convs run on the generic
    implicit-GEMM kernel, BatchNorm+LeakyReLU fused."""

    def __init__(self, dim: int = 64, in_channels: int = 1) -> None:
        super().__init__()
        d = dim
        spec = [(in_channels, d, 4, 2, 1), (d, 2 * d, 4, 2, 1), (2 * d, 4 * d, 4, 2, 1), (4 * d, 8 * d, 4, 2, 1),
                (8 * d, 16 * d, 4, 2, 1), (16 * d, 32 * d, 4, 2, 1), (32 * d, 16 * d, 1, 1, 0),
                (16 * d, 8 * d, 1, 1, 0), (8 * d, 2 * d, 1, 1, 0), (2 * d, 2 * d, 3, 1, 1), (2 * d, 8 * d, 3, 1, 1)]
        for i, (ci, co, k, s, p) in enumerate(spec, 1):
            setattr(self, f"conv{i}", Conv2d(ci, co, kernel_size=k, stride=s, padding=p))
            if i > 1:
                setattr(self, f"bn{i - 1}", BatchNorm2d(co))
        self.global_avg_pool = nn.Identity()
        self.fc = Linear(8 * d, 1)
        self.activation = LeakyReLU(0.2, inplace=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.conv1(x, ACT_LEAKY)
        for i in range(2, 9):
            x = getattr(self, f"bn{i - 1}")(getattr(self, f"conv{i}")(x), ACT_LEAKY)
        res = x
        for i in range(9, 12):
            x = getattr(self, f"bn{i - 1}")(getattr(self, f"conv{i}")(x), ACT_LEAKY)
        x = ops.add(x, res)
        x = ops.global_avg_pool(x)
        return self.fc(x)


__all__ = ["SRGAND", "Discriminator1"]
