"""Oracle ``nn.Module`` containers: reference class names, ctor signatures and
``state_dict`` keys; forwards go through ``oracle.functional``.

Test infrastructure only (see ``oracle/__init__.py``).  torch.nn layer objects
are used purely as parameter holders so the key names equal the reference's
(models/generator.py, models/discriminator.py, models/losses.py, models/utils.py).
"""
from __future__ import annotations

import math
import warnings
from typing import List, Optional, Sequence

import torch
from torch import nn

from . import functional as OF


# ---- helpers applying a holder layer through the explicit arithmetic ------
def _conv(layer: nn.Conv2d, x):
    return OF.conv2d(x, layer.weight, layer.bias, stride=layer.stride[0], padding=layer.padding[0])


def _bn(layer: nn.BatchNorm2d, x):
    if layer.training:
        layer.num_batches_tracked += 1
        return OF.batch_norm_train(x, layer.weight, layer.bias, layer.running_mean, layer.running_var,
                                   layer.momentum, layer.eps)
    return OF.batch_norm_eval(x, layer.weight, layer.bias, layer.running_mean, layer.running_var, layer.eps)


class _Slot(nn.Module):
    """Parameter-free placeholder keeping nn.Sequential indices aligned with the reference."""

    def forward(self, x):  # pragma: no cover - never called
        raise RuntimeError("placeholder")


# ---- generator pieces (generator.py:29-157) --------------------------------
class DenseLayer(nn.Module):
    """generator.py:29-38: cat([x, conv3x3(relu(bn(x)))])."""

    def __init__(self, in_channels: int, growth_rate: int) -> None:
        super().__init__()
        self.bn = nn.BatchNorm2d(in_channels)
        self.relu = _Slot()
        self.conv = nn.Conv2d(in_channels, growth_rate, 3, padding=1)

    def forward(self, x):
        return torch.cat([x, _conv(self.conv, OF.relu(_bn(self.bn, x)))], 1)


class DenseBlock(nn.Module):
    """generator.py:41-54."""

    def __init__(self, num_layers: int, in_channels: int, growth_rate: int) -> None:
        super().__init__()
        self.layers = nn.ModuleList(
            DenseLayer(in_channels + i * growth_rate, growth_rate) for i in range(num_layers))

    def forward(self, x):
        for lyr in self.layers:
            x = lyr(x)
        return x


class TransitionLayer(nn.Module):
    """generator.py:57-67: BN -> ReLU -> conv1x1 (no pooling)."""

    def __init__(self, in_channels: int, out_channels: int) -> None:
        super().__init__()
        self.layer = nn.Sequential(nn.BatchNorm2d(in_channels), _Slot(), nn.Conv2d(in_channels, out_channels, 1))

    def forward(self, x):
        return _conv(self.layer[2], OF.relu(_bn(self.layer[0], x)))


class PAMModule(nn.Module):
    """generator.py:104-122."""

    def __init__(self, channels: int) -> None:
        super().__init__()
        r = max(1, channels // 8)
        self.query = nn.Conv2d(channels, r, 1)
        self.key = nn.Conv2d(channels, r, 1)
        self.value = nn.Conv2d(channels, channels, 1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return OF.pam(x, self.query.weight, self.query.bias, self.key.weight, self.key.bias,
                      self.value.weight, self.value.bias, self.gamma)


class CAMModule(nn.Module):
    """generator.py:125-139 (``channels`` unused there too)."""

    def __init__(self, channels: int) -> None:
        super().__init__()
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return OF.cam(x, self.gamma)


class DANetAttention(nn.Module):
    """generator.py:142-157: fuse(cat([PAM(x), CAM(x)]))."""

    def __init__(self, channels: int) -> None:
        super().__init__()
        self.position_attention = PAMModule(channels)
        self.channel_attention = CAMModule(channels)
        self.fuse = nn.Sequential(nn.Conv2d(2 * channels, channels, 3, padding=1, bias=False),
                                  nn.BatchNorm2d(channels), _Slot())

    def forward(self, x):
        feats = torch.cat([self.position_attention(x), self.channel_attention(x)], 1)
        return OF.relu(_bn(self.fuse[1], _conv(self.fuse[0], feats)))


def _build_attention(attention_type: Optional[str], channels: int) -> Optional[nn.Module]:
    """generator.py:160-172; the reference forgets ``import warnings`` (NameError
    on 'senet'/'cbam'); the evident intent -- alias to danet -- is kept."""
    if attention_type is None or attention_type.lower() == "none":
        return None
    kind = attention_type.lower()
    if kind in ("senet", "cbam"):
        warnings.warn(f"Attention type '{attention_type}' currently aliases to 'danet'.", RuntimeWarning)
        kind = "danet"
    if kind != "danet":
        raise ValueError(f"Unsupported attention type: {attention_type}")
    return DANetAttention(channels)


class FlexibleUpsamplingModule(nn.Module):
    """generator.py:175-247."""

    def __init__(self, input_channels: int = 40, growth_rate: int = 24, num_blocks: int = 3,
                 num_layers_per_block: int = 4, attention_type: Optional[str] = "danet") -> None:
        super().__init__()
        self.initial = nn.Sequential(nn.Conv2d(input_channels, 64, 3, padding=1, bias=False),
                                     nn.BatchNorm2d(64), _Slot())
        self.dense_blocks = nn.ModuleList()
        self.transition_layers = nn.ModuleList()
        self.attention_modules = nn.ModuleList()
        self.feature_channels: List[int] = []
        width = 64
        for i in range(num_blocks):
            self.dense_blocks.append(DenseBlock(num_layers_per_block, width, growth_rate))
            width += num_layers_per_block * growth_rate
            self.attention_modules.append(_build_attention(attention_type, width))
            self.feature_channels.append(width)
            if i + 1 < num_blocks:
                self.transition_layers.append(TransitionLayer(width, width // 2))
                width //= 2
        self.channel_adjust = nn.ModuleList(
            nn.Conv2d(ch, 64, 1, bias=False) for ch in self.feature_channels[::-1])
        self.upsample = nn.Sequential(
            nn.Conv2d(width, 64, 3, padding=1, bias=False), nn.BatchNorm2d(64), _Slot(), _Slot(),
            nn.Conv2d(64, 64, 3, padding=1, bias=False), nn.BatchNorm2d(64), _Slot(), _Slot())
        self.final = nn.Conv2d(64, 1, 3, padding=1)

    def forward(self, x):
        x = OF.relu(_bn(self.initial[1], _conv(self.initial[0], x)))
        skips = []
        for i, blk in enumerate(self.dense_blocks):
            x = blk(x)
            att = self.attention_modules[i]
            if att is not None:
                x = att(x)
            skips.append(x)
            if i < len(self.transition_layers):
                x = self.transition_layers[i](x)
        for conv_i, bn_i in ((0, 1), (4, 5)):
            x = OF.relu(_bn(self.upsample[bn_i], _conv(self.upsample[conv_i], x)))
            x = OF.bicubic_resize(x, 2 * x.shape[2], 2 * x.shape[3], 0.5, 0.5)
        for adjust, feat in zip(self.channel_adjust, skips[::-1]):
            x = x + _conv(adjust, OF.bilinear_resize(feat, x.shape[2], x.shape[3]))
        return _conv(self.final, x)


# ---- exported-but-unused modules (generator.py:11-26,70-101) ---------------
class OriginalRelationshipLearner(nn.Module):
    def __init__(self, input_channels: int) -> None:
        super().__init__()
        mods, cin = [], input_channels
        for cout in (64, 128, 256, 512, 1024):
            mods += [nn.Conv2d(cin, cout, 3, padding=1), _Slot()]
            cin = cout
        self.net = nn.Sequential(*mods)

    def forward(self, x):
        for i in range(0, len(self.net), 2):
            x = OF.relu(_conv(self.net[i], x))
        return x


class SqueezeExcitation(nn.Module):
    def __init__(self, channels: int, reduction_ratio: int = 16) -> None:
        super().__init__()
        red = max(1, channels // reduction_ratio)
        self.fc1 = nn.Conv2d(channels, red, 1)
        self.fc2 = nn.Conv2d(red, channels, 1)

    def forward(self, x):
        a = x.mean(dim=(2, 3), keepdim=True)
        a = torch.sigmoid(_conv(self.fc2, OF.relu(_conv(self.fc1, a))))
        return x * a


class CBAMBlock(nn.Module):
    def __init__(self, channels: int, reduction_ratio: int = 16) -> None:
        super().__init__()
        self.channel_attention = SqueezeExcitation(channels, reduction_ratio)
        self.spatial_attention = nn.Sequential(nn.Conv2d(2, 1, 7, padding=3, bias=False), _Slot())

    def forward(self, x):
        x = self.channel_attention(x)
        a = torch.cat([x.amax(dim=1, keepdim=True), x.mean(dim=1, keepdim=True)], 1)
        return x * torch.sigmoid(_conv(self.spatial_attention[0], a))


# ---- discriminators (discriminator.py) -------------------------------------
class Discriminator1(nn.Module):
    """discriminator.py:57-77.  fc1 is lazy there; here it materialises on the
    first forward with nn.Linear's default init (what LazyLinear does)."""

    def __init__(self, input_channels: int = 1) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(input_channels, 64, 3, stride=2, padding=1)
        self.conv2 = nn.Conv2d(64, 128, 3, stride=2, padding=1)
        self.conv3 = nn.Conv2d(128, 256, 3, stride=2, padding=1)
        self.conv4 = nn.Conv2d(256, 512, 3, stride=2, padding=1)
        self.fc1 = nn.LazyLinear(1024)
        self.fc2 = nn.Linear(1024, 1)

    def forward(self, x):
        for c in (self.conv1, self.conv2, self.conv3, self.conv4):
            x = OF.leaky_relu(_conv(c, x), 0.2)
        x = x.flatten(1)
        if isinstance(self.fc1, nn.LazyLinear) and self.fc1.has_uninitialized_params():
            self.fc1(x)  # materialise exactly as LazyLinear would
        x = OF.leaky_relu(OF.linear(x, self.fc1.weight, self.fc1.bias), 0.2)
        return OF.linear(x, self.fc2.weight, self.fc2.bias)


class SRGAND(nn.Module):
    """discriminator.py:8-54 (exported, unused by the train loop)."""

    def __init__(self, dim: int = 64, in_channels: int = 1) -> None:
        super().__init__()
        d = dim
        spec = [(in_channels, d, 4, 2, 1), (d, 2 * d, 4, 2, 1), (2 * d, 4 * d, 4, 2, 1), (4 * d, 8 * d, 4, 2, 1),
                (8 * d, 16 * d, 4, 2, 1), (16 * d, 32 * d, 4, 2, 1), (32 * d, 16 * d, 1, 1, 0),
                (16 * d, 8 * d, 1, 1, 0), (8 * d, 2 * d, 1, 1, 0), (2 * d, 2 * d, 3, 1, 1), (2 * d, 8 * d, 3, 1, 1)]
        for i, (ci, co, k, s, p) in enumerate(spec, 1):
            setattr(self, f"conv{i}", nn.Conv2d(ci, co, k, stride=s, padding=p))
            if i > 1:
                setattr(self, f"bn{i - 1}", nn.BatchNorm2d(co))
        self.fc = nn.Linear(8 * d, 1)

    def forward(self, x):
        x = OF.leaky_relu(_conv(self.conv1, x))
        for i in range(2, 9):
            x = OF.leaky_relu(_bn(getattr(self, f"bn{i - 1}"), _conv(getattr(self, f"conv{i}"), x)))
        res = x
        for i in range(9, 12):
            x = OF.leaky_relu(_bn(getattr(self, f"bn{i - 1}"), _conv(getattr(self, f"conv{i}"), x)))
        x = (x + res).mean(dim=(2, 3))
        return OF.linear(x, self.fc.weight, self.fc.bias)


# ---- losses (losses.py) ------------------------------------------------------
class TVLoss(nn.Module):
    def __init__(self, weight: float = 1.0) -> None:
        super().__init__()
        self.weight = weight

    def forward(self, x):
        return OF.tv_loss(x, self.weight)


class SSIM(nn.Module):
    def __init__(self, window_size: int = 11, size_average: bool = True) -> None:
        super().__init__()
        self.window_size, self.size_average, self.channel = window_size, size_average, 1
        self.register_buffer("window", OF.gaussian_window(window_size)[None, None].contiguous())

    def forward(self, img1, img2):
        return OF.ssim(img1, img2, self.window_size, self.size_average)


class PerceptualLoss(nn.Module):
    """losses.py:13-73 with the random-init fallback (losses.py:42-48) as the
    only offline-possible weights; ``self.vgg`` keeps torchvision's indices so a
    VGG19 ``features`` state_dict loads (keys ``<idx>.weight``)."""

    def __init__(self, feature_layers: Sequence[int] = (1, 6, 11, 20), weights_path: Optional[str] = None,
                 pretrained: bool = True, device=None, use_gpu=None) -> None:
        super().__init__()
        self.feature_layers = set(feature_layers)
        if not self.feature_layers:
            raise ValueError("feature_layers must contain at least one index")
        if weights_path is None and pretrained:
            warnings.warn("Falling back to randomly initialised VGG19 features (no network in this image).",
                          RuntimeWarning)
        mods = []
        for ent in OF.VGG19_FEATURES_21[: max(self.feature_layers) + 1]:
            mods.append(nn.Conv2d(ent[1], ent[2], 3, padding=1) if ent[0] == "conv" else _Slot())
        self.vgg = nn.Sequential(*mods)
        if weights_path is not None:
            self.vgg.load_state_dict(torch.load(weights_path, map_location="cpu", weights_only=True), strict=False)
        self.vgg.eval()
        for p in self.vgg.parameters():
            p.requires_grad_(False)

    def forward(self, x, y):
        params = [(m.weight, m.bias) for m in self.vgg if isinstance(m, nn.Conv2d)]
        return OF.perceptual(x, y, params, sorted(self.feature_layers))


# ---- init (utils.py:7-21) ----------------------------------------------------
def weights_init_normal(module: nn.Module) -> None:
    """Kaiming-normal(fan_in, relu) conv weights, BN (1, 0), Xavier-normal linear, zero biases."""
    if isinstance(module, nn.Conv2d):
        fan_in = module.weight.shape[1] * module.weight.shape[2] * module.weight.shape[3]
        with torch.no_grad():
            module.weight.normal_(0.0, math.sqrt(2.0 / fan_in))
            if module.bias is not None:
                module.bias.zero_()
    elif isinstance(module, nn.BatchNorm2d):
        with torch.no_grad():
            module.weight.fill_(1.0)
            module.bias.zero_()
    elif isinstance(module, nn.Linear):
        fan_out, fan_in = module.weight.shape
        with torch.no_grad():
            module.weight.normal_(0.0, math.sqrt(2.0 / (fan_in + fan_out)))
            if module.bias is not None:
                module.bias.zero_()
