"""Losses of the GAN-DANet G step on HIP kernels (models/losses.py): PerceptualLoss, TVLoss, SSIM, plus the
BCE-with-logits / MSE criteria the notebook takes from torch.nn (GAN_DANet_train.ipynb:L190-191)."""
from __future__ import annotations

import warnings
from typing import Optional, Sequence, Set

import torch
from torch import nn

from . import ops
from .layers import ACT_RELU, Conv2d, ReLU

# torchvision.models.vgg19(...).features[:21] ("E" configuration): index -> layer
_VGG19_HEAD = ("c64", "r", "c64", "r", "p", "c128", "r", "c128", "r", "p",
               "c256", "r", "c256", "r", "c256", "r", "c256", "r", "p", "c512", "r")


class _MaxPool2(nn.Module):
    def forward(self, x):
        return ops.MaxPool2Fn.apply(x)


class PerceptualLoss(nn.Module):
    """losses.py:13-73: sum of L1 distances between VGG19 feature maps at ``feature_layers``; 1-channel inputs are
    repeated to 3 channels; VGG is frozen / eval.  Offline there are no pretrained weights: like the reference's
    fallback (losses.py:42-48) the features are randomly initialised unless ``weights_path`` is given.
    ``use_gpu`` is accepted because the notebook still passes it (L194)."""

    def __init__(self, feature_layers: Sequence[int] = (1, 6, 11, 20), weights_path: Optional[str] = None,
                 pretrained: bool = True, device: Optional[torch.device] = None, use_gpu: Optional[bool] = None,
                 chunk: int = 4) -> None:
        super().__init__()
        self.feature_layers: Set[int] = set(feature_layers)
        if not self.feature_layers:
            raise ValueError("feature_layers must contain at least one index")
        top = max(self.feature_layers)
        if top >= len(_VGG19_HEAD):
            raise ValueError("feature_layers beyond VGG19.features[:21] are not supported")
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.device = device
        self.chunk = chunk
        if weights_path is None and pretrained:
            warnings.warn("Falling back to randomly initialised VGG19 features (no pretrained weights offline). "
                          "Pass pretrained=False or provide weights_path to silence this warning.", RuntimeWarning)
        mods, cin = [], 3
        for ent in _VGG19_HEAD[: top + 1]:
            if ent[0] == "c":
                cout = int(ent[1:])
                mods.append(Conv2d(cin, cout, kernel_size=3, padding=1))
                cin = cout
            elif ent == "r":
                mods.append(ReLU())
            else:
                mods.append(_MaxPool2())
        self.vgg = nn.Sequential(*mods)
        if weights_path is not None:
            state = torch.load(weights_path, map_location="cpu", weights_only=True)
            missing, unexpected = self.vgg.load_state_dict(state, strict=False)
            if unexpected:
                warnings.warn(f"Unexpected keys when loading VGG weights: {unexpected}", RuntimeWarning)
            if missing:
                warnings.warn(f"Missing keys when loading VGG weights: {missing}", RuntimeWarning)
        self.vgg.to(device)
        self.vgg.eval()
        for p in self.vgg.parameters():
            p.requires_grad_(False)

    def _features(self, t: torch.Tensor):
        """walk the stack; conv+ReLU pairs run as one kernel (ReLU is in place in torchvision, so a tap at a conv
        index sees rectified values only when the NEXT index is a ReLU that has already run -- taps are at ReLU
        indices in every configuration the reference uses, which is what is supported here)."""
        feats = {}
        i, n = 0, len(self.vgg)
        while i < n:
            m = self.vgg[i]
            if isinstance(m, Conv2d):
                if i + 1 < n and isinstance(self.vgg[i + 1], ReLU) and i not in self.feature_layers:
                    t = m(t, ACT_RELU)
                    if i + 1 in self.feature_layers:
                        feats[i + 1] = t
                    i += 2
                    continue
                t = m(t)
            else:
                t = m(t)
            if i in self.feature_layers:
                feats[i] = t
            i += 1
        return feats

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        x3 = x if x.shape[1] == 3 else ops.repeat_channels(x, 3)
        y3 = y if y.shape[1] == 3 else ops.repeat_channels(y, 3)
        with torch.no_grad():
            fy = self._features(y3)
        fx = self._features(x3)
        terms = [ops.l1_loss(fx[i], fy[i]) for i in sorted(self.feature_layers)]
        return ops.weighted_sum([1.0] * len(terms), terms)


class TVLoss(nn.Module):
    """losses.py:76-87"""

    def __init__(self, weight: float = 1.0) -> None:
        super().__init__()
        self.weight = weight

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.tv_loss(x, self.weight)


class SSIM(nn.Module):
    """losses.py:90-147, forward value only (the train loop evaluates it and drops it, L263/L267).
    ``size_average=False`` is not on the path and not implemented."""

    def __init__(self, window_size: int = 11, size_average: bool = True) -> None:
        super().__init__()
        self.window_size = window_size
        self.size_average = size_average
        self.channel = 1
        coords = torch.arange(window_size, dtype=torch.float32)
        g = torch.exp(-((coords - window_size // 2) ** 2) / (2 * 1.5 ** 2))
        g = (g / g.sum()).unsqueeze(1)
        self.register_buffer("window", (g @ g.t()).unsqueeze(0).unsqueeze(0).contiguous())

    def forward(self, img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
        if not self.size_average:
            raise NotImplementedError("SSIM(size_average=False) is not on the G+D path")
        return ops.ssim_value(img1, img2, self.window_size)


class BCEWithLogitsLoss(nn.Module):
    """torch.nn.BCEWithLogitsLoss() for the all-ones / all-zeros targets the train loop uses (L249-253, L261)."""

    def forward(self, logits: torch.Tensor, target) -> torch.Tensor:
        label = float(target) if not torch.is_tensor(target) else None
        if label is None:
            raise NotImplementedError("pass the constant label (1.0 / 0.0); per-element targets are not on the path")
        return ops.bce_with_logits(logits, label)


class MSELoss(nn.Module):
    def forward(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        return ops.mse_loss(a, b)


__all__ = ["PerceptualLoss", "TVLoss", "SSIM"]
